cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "f43" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "winograd_f43" 2>&1 | tail -8
python tools/bench_wino.py 2>&1 | grep -v amdgpu.ids > $O/r5_winograd_gate.txt; cat $O/r5_winograd_gate.txt
for v in 0 1; do
DP_WINO43=$v python bench.py --config ddim --no-roofline > $O/r5_ddim_w43_$v.json 2>/dev/null
DP_WINO43=$v python bench.py --config ldm --steps 2 --warmup 1 --no-roofline > $O/r5_ldm_w43_$v.json 2>/dev/null
done
python - <<'PY'
import json
for f in ('r5_ddim_w43_0','r5_ddim_w43_1','r5_ldm_w43_0','r5_ldm_w43_1'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); print(f, round(d['value'],2), d['unit'], 'ms/step', round(d['ms_per_step'],2))
    except Exception as e: print(f,'ERR',e)
PY

cd $GRAFT_REPO_ROOT
echo "== probe"; ./tools/probe/lds_probe_x4
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or linear or bmm" 2>&1 | tail -3
for mp in 512 256 128; do echo "== C1 MIN_PIX=$mp"; DP_WGRAD_MIN_PIX=$mp python tools/bench_c1.py 2>&1 | tail -1; done
for mp in 512 256; do echo "== bedroom MIN_PIX=$mp"; DP_WGRAD_MIN_PIX=$mp python tools/bench_bedroom.py 4 2>&1 | grep bedroom; done
echo "== ldm"; python tools/bench_ldm.py 2>&1 | tail -2
echo "== secondary"; python tools/bench_secondary.py 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench4.json 2>gpurun_out/r2_bench4.err
python - <<PY
import json
b=json.load(open('gpurun_out/r2_bench4.json'))
print('bench', b['ms_per_step'], b['value'], b['config']['kernel_launches_per_step'])
PY

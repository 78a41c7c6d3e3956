cd $GRAFT_REPO_ROOT; O=gpurun_out
python -m pytest tests/test_e2e_gpu.py -q -x -k "general_cross_attention or multi_head or tiny_sweep or cifar_c1 or ldm_unet or ldm_prune or ldm_importance or bedroom_topology or autograd_bridge or two_timesteps or hipgraph" --durations=5 2>&1 | tail -8
python -m pytest tests/test_rccl_gpu.py -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/r5_bench_c.json 2> $O/r5_bench_c.err
DP_NO_FUSED_QKV_WGRAD=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/r5_bench_c_noqkv.json 2>/dev/null
python bench.py --config c4_finetune --no-roofline > $O/r5_c4_c.json 2>/dev/null
python bench.py --config bedroom256 --no-roofline --no-cpu-baseline > $O/r5_bedroom_c.json 2>/dev/null
python - <<'PY'
import json
for f in ('r5_bench_c','r5_bench_c_noqkv','r5_c4_c','r5_bedroom_c'):
    try:
        d=json.load(open('gpurun_out/%s.json'%f)); c=d['config']
        print(f, round(d['value'],1), d['unit'], 'ms/step', round(d['ms_per_step'],2), 'launches', c.get('kernel_launches_per_step'))
        r=d.get('roofline')
        if r:
            for k,v in sorted(r['kernels'].items(), key=lambda kv:-kv[1]['ms'])[:8]: print('      %-55s %4d %7.2f ms %6.1f TF/s'%(k,v['launches'],v['ms'],v['tflops']))
    except Exception as e: print(f, 'ERR', e)
PY

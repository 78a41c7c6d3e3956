cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or bmm or linear" 2>&1 | tail -8
python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "cifar_c1 or tiny_prune or pruned_model or ldm_unet or multi_head" 2>&1 | tail -4
for cfg in "A=1" "DP_NO_X4=1"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench5.json 2> gpurun_out/r2_bench5.err
  echo "== $cfg"; python - <<PY
import json
b=json.load(open('gpurun_out/r2_bench5.json'))
r=b['roofline']
print(b['ms_per_step'], b['value'], b['config']['kernel_launches_per_step'], r['kernel'], round(r['achieved'],1), {k:(v['launches'],round(v['tflops'],1),round(v['ms'],2)) for k,v in r['kernels'].items() if v['ms']>1})
PY
done
echo "== C1"; python tools/bench_c1.py 2>&1 | tail -1
echo "== bedroom"; python tools/bench_bedroom.py 4 2>&1 | grep bedroom
echo "== ldm"; python tools/bench_ldm.py 2>&1 | tail -2
echo "== secondary"; python tools/bench_secondary.py 2>&1 | tail -3

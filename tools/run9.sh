cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv" 2>&1 | tail -2
echo "== n64 on"; python tools/bench_conv_only.py
echo "== n64 off"; DP_CONV_N64=0 python tools/bench_conv_only.py
for cfg in "A=1" "DP_CONV_N64=0" "DP_CONV_N64=256,2048" "DP_CONV_N64=1000,1100"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench9.json 2> gpurun_out/r2_bench9.err
  echo "== $cfg"; python - <<PY
import json
b=json.load(open('gpurun_out/r2_bench9.json'))
r=b['roofline']
print(b['ms_per_step'], b['value'], b['config']['kernel_launches_per_step'], {k:(v['launches'],round(v['tflops'],1),round(v['ms'],2)) for k,v in r['kernels'].items() if v['ms']>1})
PY
done

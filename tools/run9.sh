cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv or xcd" 2>&1 | tail -2
echo "== spread"; python tools/bench_wgrad.py 2>&1 | grep -v amdgpu
echo "== front"; DP_HIP_LIB=$GRAFT_REPO_ROOT/diff-pruning_amd/libdp_hip_e1.so python tools/bench_wgrad.py 2>&1 | grep -v amdgpu
for cfg in "A=1" "DP_HIP_LIB=$GRAFT_REPO_ROOT/diff-pruning_amd/libdp_hip_e1.so"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench9.json 2> gpurun_out/r2_bench9.err
  echo "== $cfg"; python - <<PY
import json
b=json.load(open('gpurun_out/r2_bench9.json'))
r=b['roofline']
print(b['ms_per_step'], b['value'], {k:(v['launches'],round(v['tflops'],1),round(v['ms'],2)) for k,v in r['kernels'].items() if v['ms']>1})
PY
done

cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "winograd" 2>&1 | tail -3
for l in "" novp ""; do
  if [ -n "$l" ]; then export DP_HIP_LIB=$PWD/diff-pruning_amd/libdp_hip_$l.so; else unset DP_HIP_LIB; fi
  echo "=== lib ${l:-default(vpipe)}"
  timeout 300 python tools/bench_wino.py 2>&1 | grep -v amdgpu | grep -E "256\+0  ->256 @16|128\+0  ->128 @32|192"
done

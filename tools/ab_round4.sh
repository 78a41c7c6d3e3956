cd $GRAFT_REPO_ROOT
for l in "" a1b3 a1b2 a0b2 a0b1 a2b4 a0b0 r3 ""; do
  if [ -n "$l" ]; then export DP_HIP_LIB=$PWD/diff-pruning_amd/libdp_hip_$l.so; else unset DP_HIP_LIB; fi
  echo "=== lib ${l:-default(a1b4)}"
  python tools/bench_conv_only.py 2>&1 | grep conv
done

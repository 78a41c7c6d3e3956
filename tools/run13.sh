cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "A=1" "DP_NO_FEW_OUT=1"; do
  echo "== $cfg"; env $cfg python tools/bench_secondary.py 2>&1 | grep -v amdgpu | grep -i "finetune\|DDIM"
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -- python tools/bench_secondary.py > /dev/null 2>&1
f=$(find /tmp/prof_c4 -name '*kernel_stats.csv' | head -1); head -14 "$f" | cut -c1-150
grep -i "few_out\|64, 128" "$f" | cut -c1-150

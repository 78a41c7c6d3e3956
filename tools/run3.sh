# round-2 profiles of the secondary configs: one rocprofv3 kernel summary per config
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
prof() {  # name, command...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- "$@" > gpurun_out/r2_prof_$name.log 2>&1
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/round2_${name}_kernel_stats.csv 2>/dev/null
  grep -v "^\[" gpurun_out/r2_prof_$name.log | grep -v amdgpu.ids | tail -6
}
prof ldm python tools/bench_ldm.py
prof c4_ddim python tools/bench_secondary.py
prof bedroom python tools/bench_bedroom.py 4
prof c1 python tools/bench_c1.py
echo "== unprofiled C1 with two half pipelines"; DP_HALVES=2 python tools/bench_c1.py 2>&1 | tail -2
echo "== unprofiled C1"; python tools/bench_c1.py 2>&1 | tail -2
echo "== unprofiled bedroom halves=2"; DP_HALVES=2 python tools/bench_bedroom.py 4 2>&1 | tail -2
echo "== unprofiled bedroom"; python tools/bench_bedroom.py 4 2>&1 | tail -2

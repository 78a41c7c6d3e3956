# Round-2 evidence: GPU tests, bench line, rocprofv3 kernel summaries (default + serial), PMC traffic passes, secondary configs.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
python -m pytest tests -m gpu -q --durations=8 > $O/round2_gpu_tests.log 2>&1; tail -14 $O/round2_gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/round2_bench_line.json 2> $O/round2_bench_line.err; tail -c 600 $O/round2_bench_line.json
stats() {  # name, env..., -- command
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- "$@" > $O/r2_prof_$name.log 2>&1
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1); cp "$f" $O/round2_${name}_kernel_stats.csv 2>/dev/null
}
stats bench python bench.py --steps 10 --warmup 2 --no-cpu-baseline
DP_NO_OVERLAP=1 stats bench_serial python bench.py --steps 10 --warmup 2 --no-cpu-baseline
for ctr in FETCH_SIZE WRITE_SIZE; do
  DP_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r2_pmc_$ctr.log 2>&1
done
python tools/pmc_aggregate.py $O/round2_pmc_bench_traffic.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
( python tools/bench_secondary.py; python tools/bench_bedroom.py 4; python tools/bench_ldm.py; python tools/bench_c1.py ) 2>&1 | grep -v amdgpu.ids > $O/round2_secondary_metrics.log
cat $O/round2_secondary_metrics.log
stats ldm python tools/bench_ldm.py
stats c4_ddim python tools/bench_secondary.py
stats bedroom python tools/bench_bedroom.py 4
stats c1 python tools/bench_c1.py

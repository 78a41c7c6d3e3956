# Round-4 evidence (one MI355X via gpurun).  usage: bash tools/run_evidence.sh tests|bench|profiles|pmc
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
R=round4
stats() {  # name, -- command: rocprofv3 per-kernel summary of one bench.py configuration
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- "$@" > $O/r4_prof_$name.log 2>&1
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1); cp "$f" $O/${R}_${name}_kernel_stats.csv 2>/dev/null
}
case "$1" in
tests)
  python -m pytest tests -m gpu -q --durations=10 > $O/${R}_gpu_tests.log 2>&1; tail -16 $O/${R}_gpu_tests.log
  cp $O/test_report.json $O/${R}_test_report.json ;;
bench)
  python bench.py --steps 20 --warmup 5 > $O/${R}_bench_line.json 2> $O/${R}_bench_line.err; tail -c 400 $O/${R}_bench_line.json
  for c in bedroom256 c4_finetune ddim ldm; do
    python bench.py --config $c > $O/${R}_bench_$c.json 2> $O/${R}_bench_$c.err; python -c "
import json; d=json.load(open('$O/${R}_bench_$c.json')); r=d['roofline']
print('$c', round(d['value'],2), d['unit'], round(d['ms_per_step'],2), 'ms/step; step_frac', round(r['step_frac'],3), 'ref-eq TF/s', round(r['step_tflops_reference_equivalent'],1), 'dominant', r['kernel'], round(r['achieved'],1))"
  done
  ( python tools/bench_c1.py; python tools/exp_replay.py cifar 4 eager native ) 2>&1 | grep -v amdgpu.ids > $O/${R}_c1_latency.log; cat $O/${R}_c1_latency.log
  python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids > $O/${R}_attention_fused.txt; cat $O/${R}_attention_fused.txt
  DP_FUSED_ATTN=0 python bench.py --config ddim --no-roofline 2>/dev/null | tail -1 > $O/${R}_ddim_three_launch.json
  python bench.py --config ddim --no-roofline 2>/dev/null | tail -1 > $O/${R}_ddim_fused_attn.json
  # the Winograd gate (verdict item 5) and the same headline command on the direct kernels only
  python tools/bench_wino.py 2>&1 | grep -v amdgpu.ids > $O/${R}_winograd_gate.txt; cat $O/${R}_winograd_gate.txt
  DP_WINO=0 DP_WGRAD_WINO=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${R}_bench_line_direct_kernels.json
  python -c "
import json; a=json.load(open('$O/${R}_bench_line.json')); b=json.load(open('$O/${R}_bench_line_direct_kernels.json'))
print('headline ms/step: winograd', round(a['ms_per_step'],2), 'direct kernels only', round(b['ms_per_step'],2))" ;;
profiles)
  stats bench python bench.py --steps 10 --warmup 2 --no-cpu-baseline
  DP_NO_OVERLAP=1 DP_TIMESTEP_PIPELINES=1 stats bench_serial python bench.py --steps 10 --warmup 2 --no-cpu-baseline
  stats c4_finetune python bench.py --config c4_finetune --no-roofline
  stats ddim python bench.py --config ddim --no-roofline
  stats ldm python bench.py --config ldm --no-roofline --steps 2 --warmup 1
  stats bedroom256 python bench.py --config bedroom256 --no-roofline --no-cpu-baseline
  ls -la $O/${R}_*kernel_stats.csv ;;
pmc)
  for ctr in FETCH_SIZE WRITE_SIZE; do
    DP_NO_OVERLAP=1 DP_TIMESTEP_PIPELINES=1 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/r4_pmc_$ctr.log 2>&1
  done
  python tools/pmc_aggregate.py $O/${R}_pmc_bench_traffic.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE ;;
esac

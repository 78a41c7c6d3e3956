# Round-6 evidence (one MI355X via gpurun).  usage: bash tools/run_evidence.sh tests|bench|profiles|pmc
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
R=round6
stats() {  # name, -- command: rocprofv3 per-kernel summary of one bench.py configuration
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- "$@" > $O/r6_prof_$name.log 2>&1
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1); cp "$f" $O/${R}_${name}_kernel_stats.csv 2>/dev/null
}
pmc() {  # config suffix ('' = cifar256), counters..., -- bench arguments: one --pmc pass per counter, aggregated per kernel
  sfx=$1; shift
  ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  dirs=()
  for ctr in "${ctrs[@]}"; do
    DP_NO_OVERLAP=1 DP_TIMESTEP_PIPELINES=1 DP_FINETUNE_REPLAY=0 DP_SAMPLE_REPLAY=0 rocprofv3 --kernel-trace --pmc $ctr --output-format csv \
      -d /tmp/pmc${sfx}_$ctr -- python bench.py "$@" --no-cpu-baseline --no-roofline > $O/r6_pmc${sfx}_$ctr.log 2>&1
    dirs+=(/tmp/pmc${sfx}_$ctr)
  done
  echo "${dirs[@]}"
}
case "$1" in
tests)
  # the driver's exact command line, twice (two interpreters on one lease), then smoke() as the driver runs it
  for i in 1 2; do
    ( time python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/${R}_gpu_tests_run$i.log 2>&1; echo "rc=$?" >> $O/${R}_gpu_tests_run$i.log
    grep -E "passed|failed|rc=|CHILD KILLED" $O/${R}_gpu_tests_run$i.log | tail -4
    cp $O/test_report.json $O/${R}_test_report.json; cp $O/last_test.txt $O/${R}_last_test_run$i.txt
  done
  python3 -c "import __graft_entry__ as g; g.smoke()" > $O/${R}_smoke.log 2>&1; echo "rc=$?" >> $O/${R}_smoke.log; tail -3 $O/${R}_smoke.log ;;
bench)
  python bench.py --steps 20 --warmup 5 > $O/${R}_bench_line.json 2> $O/${R}_bench_line.err; tail -c 600 $O/${R}_bench_line.json
  for c in bedroom256 c4_finetune ddim ldm; do
    python bench.py --config $c > $O/${R}_bench_$c.json 2> $O/${R}_bench_$c.err; python -c "
import json; d=json.load(open('$O/${R}_bench_$c.json')); r=d['roofline']
print('$c', round(d['value'],2), d['unit'], round(d['ms_per_step'],2), 'ms/step; step_frac', round(r['step_frac'],3), 'in the reference arithmetic', round(r['step_frac_reference_arithmetic'],3), 'dominant', r['kernel'], round(r['achieved'],1))"
  done
  # eager against natively replayed: the finetune step and the DDIM loop (verdict item 2a), same box
  DP_FINETUNE_REPLAY=0 python bench.py --config c4_finetune --no-roofline 2>/dev/null | tail -1 > $O/${R}_c4_finetune_eager.json
  DP_SAMPLE_REPLAY=0 python bench.py --config ddim --no-roofline 2>/dev/null | tail -1 > $O/${R}_ddim_eager.json
  ( python tools/bench_c1.py; python tools/exp_replay.py cifar 4 eager native ) 2>&1 | grep -v amdgpu.ids > $O/${R}_c1_latency.log; cat $O/${R}_c1_latency.log
  DP_WINO=0 DP_WGRAD_WINO=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/${R}_bench_line_direct_kernels.json
  DP_WINO2D=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > $O/${R}_bench_line_winograd_1d_only.json
  python -c "
import json; a=json.load(open('$O/${R}_bench_line.json')); b=json.load(open('$O/${R}_bench_line_direct_kernels.json')); c=json.load(open('$O/${R}_bench_line_winograd_1d_only.json'))
print('headline ms/step: F(2x2,3x3) + F(2,3)', round(a['ms_per_step'],2), '; F(2,3) only (DP_WINO2D=0)', round(c['ms_per_step'],2), '; direct kernels only', round(b['ms_per_step'],2))" ;;
profiles)
  stats bench python bench.py --steps 10 --warmup 2 --no-cpu-baseline
  DP_NO_OVERLAP=1 DP_TIMESTEP_PIPELINES=1 stats bench_serial python bench.py --steps 10 --warmup 2 --no-cpu-baseline
  DP_FINETUNE_REPLAY=0 DP_NO_OVERLAP=1 stats c4_finetune_serial python bench.py --config c4_finetune --no-roofline
  stats c4_finetune python bench.py --config c4_finetune --no-roofline
  stats ddim python bench.py --config ddim --no-roofline
  stats ldm python bench.py --config ldm --no-roofline --steps 2 --warmup 1
  DP_NO_OVERLAP=1 stats bedroom256_serial python bench.py --config bedroom256 --no-roofline --no-cpu-baseline
  for c in pruned:128 cifar:256; do
    python tools/profile_shapes.py --config ${c%%:*} --batch ${c##*:} 2>&1 | grep -v amdgpu.ids > $O/${R}_shapes_${c%%:*}${c##*:}.txt
  done
  python tools/profile_shapes.py --config ldm --batch 12 --forward-only 2>&1 | grep -v amdgpu.ids > $O/${R}_shapes_ldm_fwd12.txt
  ls -la $O/${R}_*kernel_stats.csv ;;
pmc)
  # HBM traffic of the dominant kernels, one pass per counter, per config (roofline.traffic is only quoted from the config's own file)
  python tools/pmc_aggregate.py $O/${R}_pmc_bench_traffic.json $(pmc '' FETCH_SIZE WRITE_SIZE -- --steps 3 --warmup 1)
  python tools/pmc_aggregate.py --config c4_finetune $O/${R}_pmc_bench_traffic_c4_finetune.json $(pmc _c4 FETCH_SIZE WRITE_SIZE -- --config c4_finetune --steps 3 --warmup 1)
  python tools/pmc_aggregate.py --config ldm $O/${R}_pmc_bench_traffic_ldm.json $(pmc _ldm FETCH_SIZE WRITE_SIZE -- --config ldm --steps 1 --warmup 0)
  # matrix-pipe busy cycles of the Winograd kernels (round 3 had them for the direct kernels)
  python tools/pmc_aggregate.py $O/${R}_pmc_mfma.json $(pmc _mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- --steps 3 --warmup 1) ;;
esac

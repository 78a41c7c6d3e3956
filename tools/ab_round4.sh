cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "winograd or pack_weight_batch" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_e2e_gpu.py -q -k "winograd_on_every" 2>&1 | tail -5
echo "=== bench_wino default"; timeout 200 python tools/bench_wino.py 2>&1 | grep -v amdgpu
echo "=== bench_wino WR=1 forced"; DP_WINO_WR=1 timeout 200 python tools/bench_wino.py 2>&1 | grep -v amdgpu
for cfg in c4_finetune ddim ldm bedroom256; do
 for w in 1 0; do
  echo "=== $cfg DP_WINO=$w"
  DP_WINO=$w timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r.get('kernel'), r.get('achieved'), r.get('step_frac'), r.get('step_tflops_reference_equivalent'))"
 done
done

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_e2e_gpu.py -q -k "winograd_on_every" 2>&1 | tail -4
for cfg in cifar256 c4_finetune ddim ldm bedroom256; do
 for w in 1 0; do
  echo "=== $cfg DP_WGRAD_WINO=$w"
  DP_WGRAD_WINO=$w timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r.get('kernel'), r.get('achieved'), r.get('step_frac'), r.get('step_tflops_reference_equivalent'))"
 done
done

cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "wgrad_winograd" 2>&1 | tail -5
timeout 300 python tools/bench_wino.py 2>&1 | grep -v amdgpu | grep -E "shape|wgrad"
for w in 1 0; do
  echo "=== c4_finetune DP_WGRAD_WINO=$w"
  DP_WGRAD_WINO=$w timeout 400 python bench.py --config c4_finetune --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r.get('kernel'), r.get('achieved'), r.get('step_frac'), r.get('step_tflops_reference_equivalent'))"
done

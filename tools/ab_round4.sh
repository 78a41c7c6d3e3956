cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "conv_winograd" 2>&1 | tail -6
for cfg in ldm c4_finetune cifar256; do
 for t in 512 256 1024; do
  echo "=== $cfg DP_WINO_MIN_TILES=$t"
  DP_WINO_MIN_TILES=$t timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r.get('kernel'), r.get('achieved'), r.get('step_frac'), r.get('step_tflops_reference_equivalent'))"
 done
done

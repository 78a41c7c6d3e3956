cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -k winograd 2>&1 | tail -3
for w in 1 0; do
  echo "=== DP_WINO=$w"
  DP_WINO=$w timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cifar256', d['ms_per_step'], d['value'], d['roofline'])"
done
echo "=== golden fixtures with Winograd forced on every supported layer"
DP_WINO_MIN_TILES=0 timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "tiny_forward or tiny_sweep or tiny_prune or cifar_c1 or c1_size_1000 or ddim_sampling or full_size_determinism or pruned_model_sweep" 2>&1 | tail -8

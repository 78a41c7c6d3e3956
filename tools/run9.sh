cd $GRAFT_REPO_ROOT
python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "tiny or cifar_c1 or pruned_model or bedroom_topology or finetune or multi_head or full_size or hipgraph or dropout or ddim" 2>&1 | tail -3
for cfg in "A=1" "DP_NO_TEMB_BATCH=1"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print('$cfg', b['ms_per_step'], b['value'], b['config']['kernel_launches_per_step'])"
done
echo "== C1"; python tools/bench_c1.py 2>&1 | tail -1
echo "== secondary"; python tools/bench_secondary.py 2>&1 | grep -E "finetune|DDIM UNet"

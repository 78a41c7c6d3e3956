cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "subpixel or tiny or cifar_c1 or bedroom_topology or ldm or pruned_model or finetune or ddim or stride2 or upsample2x" 2>&1 | tail -4
for cfg in "A=1" "DP_NO_UPS_SUBPIXEL=1"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print('bench', b['ms_per_step'], b['value'], b['config']['kernel_launches_per_step'])"
  env $cfg python tools/bench_secondary.py 2>&1 | grep -v amdgpu | grep -i "finetune\|DDIM UNet"
  env $cfg python tools/bench_bedroom.py 4 2>&1 | grep -v amdgpu | grep "bedroom"
  env $cfg python tools/bench_ldm.py 2>&1 | grep -v amdgpu | tail -2
done

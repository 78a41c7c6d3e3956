cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "few_output or layernorm or ldm or transformer or conv_forward_dgrad or tiny or cifar_c1 or geglu or attention" 2>&1 | tail -4
python tools/bench_ldm.py 2>&1 | tail -4
for cfg in "A=1" "DP_NO_FEW_OUT=1"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; b=json.load(sys.stdin); print('$cfg', b['ms_per_step'], b['value'], b['config']['kernel_launches_per_step'])"
done

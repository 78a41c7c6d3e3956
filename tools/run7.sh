cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "groupnorm or tiny or cifar_c1 or pruned_model or bedroom_topology or ldm or full_size or finetune or multi_head" 2>&1 | tail -5
for cfg in "A=1" "DP_NO_FUSED_ROWS=1"; do
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench7.json 2> gpurun_out/r2_bench7.err
  echo "== $cfg"; python - <<PY
import json
b=json.load(open('gpurun_out/r2_bench7.json'))
print(b['ms_per_step'], b['value'], b['config']['kernel_launches_per_step'], b['config']['host_enqueue_ms_per_step'])
PY
done
echo "== C1"; python tools/bench_c1.py 2>&1 | tail -1
echo "== secondary"; python tools/bench_secondary.py 2>&1 | tail -3
echo "== ldm"; python tools/bench_ldm.py 2>&1 | tail -2

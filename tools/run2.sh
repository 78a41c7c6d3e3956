python -m pytest tests -m gpu -q -x -k "colsum_batch or xcd or device_side or two_half or c1_size_1000 or dropout_finetune or full_size or early_exit or hipgraph or ddpm_exp or c3_bedroom" > gpurun_out/r2_tests2.log 2>&1; tail -25 gpurun_out/r2_tests2.log
i=0
for cfg in "A=1" "DP_HALVES=1" "DP_HALVES=1 DP_NO_XCD=1" "DP_HALVES=1 DP_NO_XCD=1 DP_NO_COLSUM_BATCH=1"; do
  i=$((i+1))
  env $cfg python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench2_$i.json 2> gpurun_out/r2_bench2_$i.err
  echo "== $cfg"; python - <<PY
import json
b=json.load(open('gpurun_out/r2_bench2_$i.json'))
r=b['roofline']
print(b['ms_per_step'], b['value'], b['config']['kernel_launches_per_step'], b['config']['host_enqueue_ms_per_step'], b['config'].get('half_batch_pipelines'), r['kernel'], round(r['achieved'],1), {k:(v['launches'],round(v['tflops'],1),round(v['ms'],2)) for k,v in r['kernels'].items() if v['ms']>1})
PY
done

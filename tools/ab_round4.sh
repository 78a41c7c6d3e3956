set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "splitk_fold" 2>&1 | tail -5
for mode in max4 max16 off; do
  case $mode in
    max4) export DP_SPLITK_FOLD=1 DP_SPLITK_FOLD_MAX=4;;
    max16) export DP_SPLITK_FOLD=1 DP_SPLITK_FOLD_MAX=16;;
    off) export DP_SPLITK_FOLD=0;;
  esac
  echo "=== $mode"
  timeout 400 python bench.py --config ldm --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ldm', d['ms_per_step'], d['value'])"
  timeout 300 python tools/bench_c1_long.py 2>&1 | tail -2
done

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_e2e_gpu.py -m gpu -q -k traced 2>&1 | tail -3
for sk in 512 256 1024; do
  echo "== DP_CONV_SPLITK_BLOCKS=$sk"
  DP_CONV_SPLITK_BLOCKS=$sk python tools/bench_ldm.py 2>&1 | grep -v amdgpu | tail -2
  DP_CONV_SPLITK_BLOCKS=$sk python tools/bench_secondary.py 2>&1 | grep -v amdgpu | grep -i "finetune\|DDIM UNet"
done

cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "tiny or early_exit or c1 or hipgraph or device_side" 2>&1 | tail -1; done
python tools/bench_c1_long.py 200 2>&1 | grep -v amdgpu

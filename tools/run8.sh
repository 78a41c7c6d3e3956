cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "row_sums" 2>&1 | grep -E "Error|assert|^E " | head -20

#!/bin/bash
# Build libdp_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")/diff-pruning_amd"
SRCS="csrc/gemm.hip csrc/norm.hip csrc/elementwise.hip csrc/importance.hip csrc/optim.hip csrc/transformer.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I../include -o libdp_hip.so $SRCS
echo "built $(pwd)/libdp_hip.so"

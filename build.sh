#!/bin/bash
# Build libdp_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")/diff-pruning_amd"
SRCS="csrc/gemm.hip csrc/norm.hip csrc/elementwise.hip csrc/importance.hip csrc/optim.hip csrc/transformer.hip csrc/replay.hip csrc/attention.hip csrc/winograd.hip csrc/winograd43.hip csrc/winograd2d.hip csrc/wgrad2d.hip"
# DP_EXTRA_FLAGS / DP_OUT: experiment builds (e.g. DP_EXTRA_FLAGS=-DDP_SCHED_PIPE DP_OUT=libdp_hip_exp.so, run with DP_HIP_LIB=...)
OUT="${DP_OUT:-libdp_hip.so}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I../include $DP_EXTRA_FLAGS -o "$OUT" $SRCS
echo "built $(pwd)/$OUT"

#!/usr/bin/env python3
"""Shader clock while conv_gemm runs (experiment build: DP_EXTRA_FLAGS=-DDP_CLOCK_PROBE DP_OUT=libdp_hip_exp.so ./build.sh;
DP_HIP_LIB=.../libdp_hip_exp.so python tools/clock_probe.py).  clock64 counts shader cycles, wall_clock64 100 MHz ticks."""
import ctypes as C, importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
lib = ops._lib()
B = 256
for (ci, co, h) in [(256, 256, 16), (128, 128, 32)]:
    x = torch.randn(B, ci, h, h, device='cuda')
    w = torch.randn(co, ci, 3, 3, device='cuda') / math.sqrt(ci * 9)
    wp, ld = ops.pack_weight(w, 0)
    y = torch.empty(B, co, h, h, device='cuda')
    spec = ops.ConvSpec(3, 1, 1, 0)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(30):       # sustained load: the clock of the LAST launch is read
        if rep == 20:
            s.record()
        ops.conv_forward(x, None, wp, ld, co, spec, out=y)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    out = (C.c_ulonglong * 2)()
    lib.dp_debug_read_clock.argtypes = [C.c_void_p]
    assert lib.dp_debug_read_clock(out) == 0
    fl = 2.0 * B * h * h * ci * co * 9
    print('conv %dx%d h%d: %.3f ms %.1f TFLOP/s; workgroup 0: %d shader cycles / %d wall ticks -> %.0f MHz, block life %.1f us'
          % (ci, co, h, ms, fl / ms / 1e9, out[0], out[1], 100.0 * out[0] / out[1], out[1] / 100.0))

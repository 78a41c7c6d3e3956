"""ORACLE (test infrastructure only) -- CPU restatement of the dropout masks of the finetune step.

The reference draws its masks from torch's global RNG inside nn.Dropout (diffusers/models/resnet.py:628,
attention_processor.py:457; p set by utils.set_dropout, utils.py:26-29) -- a stream no other backend can
reproduce.  The product defines its masks as a pure function of (seed, layer, step, element) through
Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
multipliers / Weyl constants) and this file restates that function in numpy, so that
  * the HIP masks are checked bit-for-bit (tests/test_kernels_gpu.py), and
  * the oracle UNet applies the SAME masks as the HIP engine when finetune steps are compared.
Pinned by the published known-answer vectors of Philox4x32-10 (tests/test_cpu.py::test_philox_known_answers).
Never imported by the product path.

Mask definition (csrc/dp_common.h): counter = (idx4 lo, idx4 hi, site, step) with idx4 = idx >> 2, key = (seed lo,
seed hi); element idx takes output word idx & 3; keep iff (word >> 8) >= ceil(p * 2^24); kept values are scaled by
1 / (1 - p) in fp32.  idx = logical index ((n_global * C + c) * HW + hw).
"""
import math
import zlib

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """c*: uint32 arrays (broadcastable), k*: python ints.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(c0, c1, c2, c3)]
    for _ in range(10):
        p0 = _M0 * c0.astype(np.uint64)
        p1 = _M1 * c2.astype(np.uint64)
        hi0, lo0 = (p0 >> _S32).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
        hi1, lo1 = (p1 >> _S32).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint32(k0), lo1, hi0 ^ c3 ^ np.uint32(k1), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def site_id(site):
    return (zlib.crc32(site.encode()) if isinstance(site, str) else int(site)) & 0xFFFFFFFF


def thr24(p):
    return int(math.ceil(p * (1 << 24)))


def dropout_multipliers(n, p, seed, site, step, idx0=0):
    """fp32 multipliers (0 or 1/(1-p)) of the logical elements [idx0, idx0 + n)."""
    idx = np.arange(idx0, idx0 + n, dtype=np.uint64)
    q = idx >> np.uint64(2)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    words = philox4x32_10((q & _MASK).astype(np.uint32), (q >> _S32).astype(np.uint32), np.uint32(site_id(site)),
                          np.uint32(int(step) & 0xFFFFFFFF), seed & 0xFFFFFFFF, seed >> 32)
    w = np.stack(words, axis=-1)                       # [n, 4]
    sel = w[np.arange(n), (idx & np.uint64(3)).astype(np.int64)]
    keep = (sel >> np.uint32(8)) >= np.uint32(thr24(p))
    return np.where(keep, np.float32(1.0 / (1.0 - p)), np.float32(0.0)).astype(np.float32)


class DropSpec:
    """What the oracle UNet needs to reproduce the engine's masks: {module name: p}, seed, step, first global image."""

    def __init__(self, table, seed=0, step=0, n_off=0):
        self.table, self.seed, self.step, self.n_off = dict(table), seed, step, n_off

    def apply(self, site, x):
        """x: torch tensor [N, C, H, W] (or [N, C, T]) -> x * mask for module `site` (identity when p == 0)."""
        import torch
        p = self.table.get(site, 0.0)
        if not p:
            return x
        per = x[0].numel()
        m = dropout_multipliers(x.numel(), p, self.seed, site, self.step, self.n_off * per)
        return x * torch.from_numpy(m).view(x.shape)


def randn(n, seed, stream_id, step, idx0=0):
    """Restatement of csrc/elementwise.hip randn_philox_kernel: fp32 standard-normal draws of the logical elements
    [idx0, idx0 + n): counter (idx >> 2 lo, hi, stream_id, step), key seed; u = w * 2^-32 + 2^-33 (one fp32 fma);
    words (0, 1) and (2, 3) are Box-Muller pairs (r cos, r sin).  The device's logf / cosf / sinf differ from numpy's in the
    last bits, so the kernel is compared to this within a few ulp of the draw, not bit for bit."""
    idx = np.arange(idx0, idx0 + n, dtype=np.uint64)
    q = idx >> np.uint64(2)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    w = philox4x32_10((q & _MASK).astype(np.uint32), (q >> _S32).astype(np.uint32), np.uint32(int(stream_id) & 0xFFFFFFFF),
                      np.uint32(int(step) & 0xFFFFFFFF), seed & 0xFFFFFFFF, seed >> 32)

    def u01(x):                      # fmaf(float(w), 2^-32, 2^-33): float(w) rounds to fp32 first, the fma rounds once
        return (x.astype(np.float32).astype(np.float64) * 2.0 ** -32 + 2.0 ** -33).astype(np.float32)
    ra = np.sqrt(np.float32(-2.0) * np.log(u01(w[0])), dtype=np.float32)
    rb = np.sqrt(np.float32(-2.0) * np.log(u01(w[2])), dtype=np.float32)
    ta = np.float32(6.2831853071795865) * u01(w[1])
    tb = np.float32(6.2831853071795865) * u01(w[3])
    v = np.stack([ra * np.cos(ta), ra * np.sin(ta), rb * np.cos(tb), rb * np.sin(tb)], axis=-1).astype(np.float32)
    return v[np.arange(n), (idx & np.uint64(3)).astype(np.int64)]

"""ORACLE (test infrastructure only) -- CPU restatement of the reference's image transforms.

torchvision is absent from the reference tree and from this environment, so the transforms are restated from their
definitions and the parity of this file is UNPINNED by reference outputs (they are three lines of arithmetic):
  torchvision.transforms.ToTensor               uint8 HWC -> fp32 CHW, x / 255
  torchvision.transforms.RandomHorizontalFlip   reverse the W axis with probability p
  torchvision.transforms.Normalize(0.5, 0.5)    (x - 0.5) / 0.5
  ddpm_exp/datasets/__init__.py:176-192         data_transform: x / 256 * 255 + u / 256 (uniform dequantization), 2 x - 1
used by utils.py:31-58 (get_dataset) and ddpm_exp/datasets/__init__.py:30-60.  The flip decisions / noise are the product's
Philox streams (oracle/philox_ref.py); never imported by the product path.
"""
import numpy as np
import torch

from . import philox_ref


def flip_decisions(n, p, seed, epoch, n_off=0, site=0xF11B):
    idx = np.arange(n_off, n_off + n, dtype=np.uint64)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    w0, _, _, _ = philox_ref.philox4x32_10((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32),
                                           np.uint32(site), np.uint32(int(epoch) & 0xFFFFFFFF), seed & 0xFFFFFFFF, seed >> 32)
    return (w0 >> np.uint32(8)) < np.uint32(philox_ref.thr24(p))


def dequant_noise(shape, seed, epoch, n_off=0, site=0xF11B):
    n = int(np.prod(shape))
    per = n // shape[0]
    idx = np.arange(n_off * per, n_off * per + n, dtype=np.uint64)
    q = idx >> np.uint64(2)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    words = philox_ref.philox4x32_10((q & np.uint64(0xFFFFFFFF)).astype(np.uint32), (q >> np.uint64(32)).astype(np.uint32),
                                     np.uint32(site ^ 0x9E3779B9), np.uint32(int(epoch) & 0xFFFFFFFF), seed & 0xFFFFFFFF, seed >> 32)
    w = np.stack(words, -1)[np.arange(n), (idx & np.uint64(3)).astype(np.int64)]
    return ((w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(shape)


def transform_batch(u8, hwc, mode=1, flip_p=0.5, seed=0, epoch=0, n_off=0, dequant=False):
    """uint8 [N,H,W,C] or [N,C,H,W] -> fp32 [N,C,H,W], fp32 arithmetic in the reference's operation order."""
    x = torch.as_tensor(np.asarray(u8))
    if hwc:
        x = x.permute(0, 3, 1, 2)
    x = x.contiguous().float().div(255)                              # ToTensor
    if flip_p:
        f = torch.from_numpy(flip_decisions(x.shape[0], flip_p, seed, epoch, n_off))
        x = torch.where(f[:, None, None, None], x.flip(-1), x)       # RandomHorizontalFlip
    if dequant:
        u = torch.from_numpy(dequant_noise(tuple(x.shape), seed, epoch, n_off))
        x = x / 256.0 * 255.0 + u / 256.0
    if mode == 1:
        x = (x - 0.5) / 0.5                                          # Normalize(mean=0.5, std=0.5)
    elif mode == 2:
        x = 2 * x - 1.0
    return x

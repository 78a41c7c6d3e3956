"""Synthetic workloads: the model configurations BASELINE.json names and seeded weights / inputs for them.

There is no network for checkpoints or datasets, so `bench.py`, `__graft_entry__.smoke()`, the tools and the golden fixtures all
run on random-init weights of the named architectures and on synthetic images.  Everything is regenerated from
`numpy.random.default_rng` streams keyed by (name, seed): the same tensors exist in the build container (where the reference
is imported to produce expected outputs) and on the GPU box (where only this repository exists).
"""
import zlib
import numpy as np
import torch

# tools/ddpm_cifar10_config.json of the reference (the authoritative CIFAR-10 UNet spec; data)
CIFAR_CFG = dict(
    sample_size=32, in_channels=3, out_channels=3, center_input_sample=False, time_embedding_type="positional",
    freq_shift=1, flip_sin_to_cos=False,
    down_block_types=["DownBlock2D", "AttnDownBlock2D", "DownBlock2D", "DownBlock2D"],
    up_block_types=["UpBlock2D", "UpBlock2D", "AttnUpBlock2D", "UpBlock2D"],
    block_out_channels=[128, 256, 256, 256], layers_per_block=2, mid_block_scale_factor=1, downsample_padding=0,
    act_fn="silu", attention_head_dim=None, norm_num_groups=32, norm_eps=1e-6)

# same topology, reduced width / resolution: small enough for full-tensor fixtures
TINY_CFG = dict(CIFAR_CFG, sample_size=16, block_out_channels=[32, 64, 64, 64], norm_num_groups=8)

# bedroom/church-256 (ddpm_exp/configs/bedroom.yml:13-20 expressed as a Diffusers config)
BEDROOM_CFG = dict(
    CIFAR_CFG, sample_size=256,
    down_block_types=["DownBlock2D", "DownBlock2D", "DownBlock2D", "DownBlock2D", "AttnDownBlock2D", "DownBlock2D"],
    up_block_types=["UpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D"],
    block_out_channels=[128, 128, 256, 256, 512, 512])


# ldm_exp/configs/latent-diffusion/cin256-v2.yaml unet_config: 400.9 M parameters
LDM_CIN256_CFG = dict(image_size=64, in_channels=3, out_channels=3, model_channels=192, attention_resolutions=[8, 4, 2],
                      num_res_blocks=2, channel_mult=[1, 2, 3, 5], num_heads=1, use_spatial_transformer=True,
                      transformer_depth=1, context_dim=512)
# same topology at reduced width / resolution for full-tensor fixtures
LDM_TINY_CFG = dict(LDM_CIN256_CFG, image_size=16, model_channels=32, context_dim=16)


def _rng(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def det_param(name, shape, seed):
    """Deterministic fp32 value for a parameter `name` (platform independent)."""
    r = _rng(name, seed)
    shape = tuple(shape)
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return (r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
    if name.endswith('weight'):      # GroupNorm gamma
        return (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
    return (0.05 * r.standard_normal(shape)).astype(np.float32)


@torch.no_grad()
def det_init_(module, seed):
    for n, p in module.named_parameters():
        p.copy_(torch.from_numpy(det_param(n, p.shape, seed)))
    return module


def det_noise(shape, seed):
    return _rng('noise', seed).standard_normal(tuple(shape)).astype(np.float32)


def det_clean(shape, seed):
    return np.clip(_rng('clean', seed).standard_normal(tuple(shape)), -1.0, 1.0).astype(np.float32)

"""FID / SSIM evaluation on the device: closes the prune -> finetune -> sample -> score loop.

  fid_score.py:100-180   get_activations                 -> InceptionV3.forward on the HIP kernels, FeatureStats.update
  fid_score.py:239-262   calculate_activation_statistics -> FeatureStats (mean / covariance accumulated on the device)
  fid_score.py:182-236   calculate_frechet_distance      -> calculate_frechet_distance (host, scipy.linalg.sqrtm, like the reference)
  fid_score.py:264-322   compute_statistics_of_path / calculate_fid_given_paths / save_fid_stats
  inception.py:16-340    InceptionV3 wrapper (blocks 0..3, resize to 299, 2x - 1) around pytorch-fid's FID Inception
  ddpm_exp/compute_ssim.py:14-53   SSIM + MSE of two sample directories -> ssim(), mse_per_image(), compare_directories()

The FID network is torchvision's Inception3 with pytorch-fid's patched pooling (inception.py:224-340).  torchvision is a
third-party dependency that is absent from the reference tree and from this environment (requirements.txt names it un-pinned);
its published architecture (BasicConv2d = bias-free conv + BatchNorm(eps 1e-3) + ReLU; InceptionA/B/C/D/E) is restated here
with the SAME module / state-dict names, so the real `pt_inception-2015-12-05-6726825d.pth` loads with `load_state_dict`.
Those weights are not available offline: tests use seeded weights and compare with the oracle's PyTorch restatement
(PARITY UNPINNED for the network, see oracle/metrics_ref.py; the Frechet distance is pinned against fid_score.py itself).
Eval-mode BatchNorm is folded into the convolution at bind time (w' = w * gamma / sqrt(var + eps), b' = beta - mean * ...), every
BasicConv2d is ONE conv_gemm launch with a bias + ReLU epilogue, and the branch outputs are written straight into channel slices
of the block's output (no torch.cat).
"""
import os
import pathlib

import numpy as np
import torch
import torch.nn as nn

from . import ops, _lib as L

IMAGE_EXTENSIONS = {'bmp', 'jpg', 'jpeg', 'pgm', 'png', 'ppm', 'tif', 'tiff', 'webp'}


# ---- parameter holders named as torchvision.models.inception ------------------------------------------------------------
class BasicConv2d(nn.Module):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        pd = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.conv = nn.Conv2d(cin, cout, ks, stride=stride, padding=pd, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)
        self.ks, self.stride, self.pd = ks, stride, pd


def _inception_a(cin, pool_features):
    m = nn.Module()
    m.branch1x1 = BasicConv2d(cin, 64, 1)
    m.branch5x5_1 = BasicConv2d(cin, 48, 1)
    m.branch5x5_2 = BasicConv2d(48, 64, 5, padding=2)
    m.branch3x3dbl_1 = BasicConv2d(cin, 64, 1)
    m.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
    m.branch3x3dbl_3 = BasicConv2d(96, 96, 3, padding=1)
    m.branch_pool = BasicConv2d(cin, pool_features, 1)
    m.kind, m.cout = 'A', 64 + 64 + 96 + pool_features
    return m


def _inception_b(cin):
    m = nn.Module()
    m.branch3x3 = BasicConv2d(cin, 384, 3, stride=2)
    m.branch3x3dbl_1 = BasicConv2d(cin, 64, 1)
    m.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
    m.branch3x3dbl_3 = BasicConv2d(96, 96, 3, stride=2)
    m.kind, m.cout = 'B', 384 + 96 + cin
    return m


def _inception_c(cin, c7):
    m = nn.Module()
    m.branch1x1 = BasicConv2d(cin, 192, 1)
    m.branch7x7_1 = BasicConv2d(cin, c7, 1)
    m.branch7x7_2 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
    m.branch7x7_3 = BasicConv2d(c7, 192, (7, 1), padding=(3, 0))
    m.branch7x7dbl_1 = BasicConv2d(cin, c7, 1)
    m.branch7x7dbl_2 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
    m.branch7x7dbl_3 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
    m.branch7x7dbl_4 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
    m.branch7x7dbl_5 = BasicConv2d(c7, 192, (1, 7), padding=(0, 3))
    m.branch_pool = BasicConv2d(cin, 192, 1)
    m.kind, m.cout = 'C', 768
    return m


def _inception_d(cin):
    m = nn.Module()
    m.branch3x3_1 = BasicConv2d(cin, 192, 1)
    m.branch3x3_2 = BasicConv2d(192, 320, 3, stride=2)
    m.branch7x7x3_1 = BasicConv2d(cin, 192, 1)
    m.branch7x7x3_2 = BasicConv2d(192, 192, (1, 7), padding=(0, 3))
    m.branch7x7x3_3 = BasicConv2d(192, 192, (7, 1), padding=(3, 0))
    m.branch7x7x3_4 = BasicConv2d(192, 192, 3, stride=2)
    m.kind, m.cout = 'D', 320 + 192 + cin
    return m


def _inception_e(cin, pool):
    m = nn.Module()
    m.branch1x1 = BasicConv2d(cin, 320, 1)
    m.branch3x3_1 = BasicConv2d(cin, 384, 1)
    m.branch3x3_2a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
    m.branch3x3_2b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
    m.branch3x3dbl_1 = BasicConv2d(cin, 448, 1)
    m.branch3x3dbl_2 = BasicConv2d(448, 384, 3, padding=1)
    m.branch3x3dbl_3a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
    m.branch3x3dbl_3b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
    m.branch_pool = BasicConv2d(cin, 192, 1)
    m.kind, m.cout, m.pool = 'E', 2048, pool          # pool: 'avg' (Mixed_7b) or 'max' (Mixed_7c, inception.py:328-333)
    return m


class FIDInception3(nn.Module):
    """torchvision Inception3(num_classes=1008, aux_logits=False) with pytorch-fid's patched blocks (inception.py:197-221):
    the module tree whose state dict the FID weight file is keyed by."""

    def __init__(self):
        super().__init__()
        self.Conv2d_1a_3x3 = BasicConv2d(3, 32, 3, stride=2)
        self.Conv2d_2a_3x3 = BasicConv2d(32, 32, 3)
        self.Conv2d_2b_3x3 = BasicConv2d(32, 64, 3, padding=1)
        self.Conv2d_3b_1x1 = BasicConv2d(64, 80, 1)
        self.Conv2d_4a_3x3 = BasicConv2d(80, 192, 3)
        self.Mixed_5b = _inception_a(192, 32)
        self.Mixed_5c = _inception_a(256, 64)
        self.Mixed_5d = _inception_a(288, 64)
        self.Mixed_6a = _inception_b(288)
        self.Mixed_6b = _inception_c(768, 128)
        self.Mixed_6c = _inception_c(768, 160)
        self.Mixed_6d = _inception_c(768, 160)
        self.Mixed_6e = _inception_c(768, 192)
        self.Mixed_7a = _inception_d(768)
        self.Mixed_7b = _inception_e(1280, 'avg')
        self.Mixed_7c = _inception_e(2048, 'max')
        self.fc = nn.Linear(2048, 1008)


def pool2d(x, k, stride, pad, mode):
    s = ops._chk_act(x)
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = ops.empty_act((N, C, Ho, Wo), x.device)
    L.check(L.load().dp_pool2d(ops._p(x), s, N, C, H, W, k, stride, pad, 0 if mode == 'max' else 1, ops._p(out), out.stride(0),
                               ops._stream()), 'dp_pool2d')
    return out


def resize_bilinear(x, size, a=1.0, b=0.0):
    s = ops._chk_act(x)
    N, C, H, W = x.shape
    out = ops.empty_act((N, C, size[0], size[1]), x.device)
    L.check(L.load().dp_resize_bilinear(ops._p(x), s, N, C, H, W, size[0], size[1], float(a), float(b), ops._p(out),
                                        ops._stream()), 'dp_resize_bilinear')
    return out


class InceptionV3(nn.Module):
    """inception.py:16-163: returns the feature maps of the selected blocks; pool3 (block 3) is the FID feature."""
    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}

    def __init__(self, output_blocks=(DEFAULT_BLOCK_INDEX,), resize_input=True, normalize_input=True, requires_grad=False,
                 use_fid_inception=True, state_dict=None):
        super().__init__()
        if not use_fid_inception:
            raise NotImplementedError('only the FID Inception structure is implemented')
        self.resize_input, self.normalize_input = resize_input, normalize_input
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, 'Last possible output block index is 3'
        self.inception = FIDInception3()
        if state_dict is not None:
            self.inception.load_state_dict(state_dict)
        for p in self.parameters():
            p.requires_grad = requires_grad
        self._packs = None

    # ---- engine ---------------------------------------------------------------------------------------------------
    def _bind(self):
        """Fold eval-mode BatchNorm into each convolution and pack the operands (once: the FID network is frozen)."""
        if self._packs is not None:
            return self._packs
        packs = {}
        for name, m in self.inception.named_modules():
            if isinstance(m, BasicConv2d):
                w, bn = m.conv.weight.detach().float(), m.bn
                scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                wf = (w * scale[:, None, None, None]).contiguous()
                bias = (bn.bias.detach().float() - bn.running_mean.detach().float() * scale).contiguous()
                wp, ld = ops.pack_weight(wf, 0)
                spec = ops.ConvSpec.general(m.ks[0], m.ks[1], m.stride, m.pd[0], m.pd[1])
                packs[name] = (wp, ld, bias, spec, w.shape[0])
        self._packs = packs
        return packs

    def _conv(self, name, x, out=None):
        wp, ld, bias, spec, cout = self._packs[name]
        return ops.conv_forward(x, None, wp, ld, cout, spec, bias=bias, out=out, relu=True)

    def _block(self, name, m, x):
        N, _, H, W = x.shape
        c = self._conv
        if m.kind in ('B', 'D'):
            Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        else:
            Ho, Wo = H, W
        out = ops.empty_act((N, m.cout, Ho, Wo), x.device)
        o = 0

        def put(t_name, src, width):
            nonlocal o
            c(name + '.' + t_name, src, out=out[:, o:o + width])
            o += width
        if m.kind == 'A':
            put('branch1x1', x, 64)
            put('branch5x5_2', c(name + '.branch5x5_1', x), 64)
            put('branch3x3dbl_3', c(name + '.branch3x3dbl_2', c(name + '.branch3x3dbl_1', x)), 96)
            put('branch_pool', pool2d(x, 3, 1, 1, 'avg'), m.cout - 224)
        elif m.kind == 'B':
            put('branch3x3', x, 384)
            put('branch3x3dbl_3', c(name + '.branch3x3dbl_2', c(name + '.branch3x3dbl_1', x)), 96)
            ops.copy_strided(pool2d(x, 3, 2, 0, 'max'), out[:, o:])
        elif m.kind == 'C':
            put('branch1x1', x, 192)
            put('branch7x7_3', c(name + '.branch7x7_2', c(name + '.branch7x7_1', x)), 192)
            t = c(name + '.branch7x7dbl_2', c(name + '.branch7x7dbl_1', x))
            put('branch7x7dbl_5', c(name + '.branch7x7dbl_4', c(name + '.branch7x7dbl_3', t)), 192)
            put('branch_pool', pool2d(x, 3, 1, 1, 'avg'), 192)
        elif m.kind == 'D':
            put('branch3x3_2', c(name + '.branch3x3_1', x), 320)
            t = c(name + '.branch7x7x3_3', c(name + '.branch7x7x3_2', c(name + '.branch7x7x3_1', x)))
            put('branch7x7x3_4', t, 192)
            ops.copy_strided(pool2d(x, 3, 2, 0, 'max'), out[:, o:])
        else:
            put('branch1x1', x, 320)
            t = c(name + '.branch3x3_1', x)
            put('branch3x3_2a', t, 384)
            put('branch3x3_2b', t, 384)
            t = c(name + '.branch3x3dbl_2', c(name + '.branch3x3dbl_1', x))
            put('branch3x3dbl_3a', t, 384)
            put('branch3x3dbl_3b', t, 384)
            put('branch_pool', pool2d(x, 3, 1, 1, m.pool), 192)
        return out

    @torch.no_grad()
    def forward(self, inp):
        if inp.device.type != 'cuda':
            raise RuntimeError('InceptionV3 runs on the MI355X HIP kernels only (no CPU / PyTorch fallback)')
        self._bind()
        x = inp.to(torch.float32).contiguous()
        a, b = (2.0, -1.0) if self.normalize_input else (1.0, 0.0)
        if self.resize_input:
            x = resize_bilinear(x, (299, 299), a, b)
        elif self.normalize_input:
            x = ops.axpby(x, 2.0, torch.full_like(x, -0.5), 2.0)          # 2x - 1
        outp = []
        c = self._conv
        x = pool2d(c('Conv2d_2b_3x3', c('Conv2d_2a_3x3', c('Conv2d_1a_3x3', x))), 3, 2, 0, 'max')
        if 0 in self.output_blocks:
            outp.append(x)
        if self.last_needed_block >= 1:
            x = pool2d(c('Conv2d_4a_3x3', c('Conv2d_3b_1x1', x)), 3, 2, 0, 'max')
            if 1 in self.output_blocks:
                outp.append(x)
        if self.last_needed_block >= 2:
            for n in ('Mixed_5b', 'Mixed_5c', 'Mixed_5d', 'Mixed_6a', 'Mixed_6b', 'Mixed_6c', 'Mixed_6d', 'Mixed_6e'):
                x = self._block(n, getattr(self.inception, n), x)
            if 2 in self.output_blocks:
                outp.append(x)
        if self.last_needed_block >= 3:
            for n in ('Mixed_7a', 'Mixed_7b', 'Mixed_7c'):
                x = self._block(n, getattr(self.inception, n), x)
            N, C, H, W = x.shape
            rows = ops.rowsum_nc(x)                                       # AdaptiveAvgPool2d((1, 1))
            x = ops.axpby(rows, 1.0 / (H * W), torch.empty_like(rows), 0.0).view(N, C, 1, 1)
            outp.append(x)
        return outp


# ---- statistics + Frechet distance ------------------------------------------------------------------------------------------
class FeatureStats:
    """Streaming mean / covariance of feature rows on the device (fid_score.py:258-261: np.mean, np.cov(rowvar=False)).
    fp32 accumulation around the first batch's mean (shifted data: no catastrophic cancellation in S2 - S1 S1^T / n); the
    final mu / sigma are formed in float64 on the host, as the reference's numpy code holds them."""

    def __init__(self, dims, device):
        self.dims, self.n = dims, 0
        self.shift = None
        self.s1 = torch.zeros(dims, dtype=torch.float32, device=device)
        self.s2 = torch.zeros((dims, dims), dtype=torch.float32, device=device)

    def update(self, feats):
        f = feats.reshape(feats.shape[0], -1).to(torch.float32).contiguous()
        assert f.shape[1] == self.dims
        assert getattr(self, '_reduced', None) is None, 'FeatureStats.update after all_reduce'
        if self.shift is None:
            k = torch.zeros(self.dims, dtype=torch.float32, device=f.device)
            ops.colsum_accum(f, f.shape[0], self.dims, 1, 0, k, False)
            self.shift = ops.axpby(k, 1.0 / f.shape[0], torch.empty_like(k), 0.0)
        c = ops.add_rowvec(f.t().contiguous().view(1, self.dims, 1, f.shape[0]), -self.shift.view(1, self.dims))
        c = c.view(self.dims, f.shape[0]).t().contiguous()                 # [n, dims] centred on the shift
        ops.colsum_accum(c, c.shape[0], self.dims, 1, 0, self.s1, True)
        ops.bmm_tn(c.unsqueeze(0), c.unsqueeze(0), out=self.s2.unsqueeze(0), accumulate=True)
        self.n += f.shape[0]

    def _host_sums(self):
        """(n, shift, s1, s2) in float64 on the host; an empty accumulator has shift 0."""
        if getattr(self, '_reduced', None) is not None:
            return self._reduced
        if self.shift is None:
            z = np.zeros(self.dims)
            return 0, z, z.copy(), np.zeros((self.dims, self.dims))
        return self.n, self.shift.double().cpu().numpy(), self.s1.double().cpu().numpy(), self.s2.double().cpu().numpy()

    def all_reduce(self, group=None):
        """Combine the accumulators of all ranks into the statistics of the union of their samples without gathering a
        single feature row: every rank re-centres its shifted sums (n, S1 = sum (x - k_r), S2 = sum (x - k_r)(x - k_r)^T) on
        rank 0's shift k in float64 -- with d = k_r - k: S1' = S1 + n d, S2' = S2 + d S1^T + S1 d^T + n d d^T -- and the
        three quantities are summed over the ranks (ONE all-reduce of 1 + dims + dims^2 doubles: 33.6 MB for dims 2048).
        Afterwards finalize() returns the same (mu, sigma) on every rank.  A no-op outside a process group."""
        import torch.distributed as dist
        from .sweep import dist_active
        if not dist_active(group):
            return self
        n, k_r, s1, s2 = self._host_sums()
        dev = self.s1.device if dist.get_backend(group) == 'nccl' else torch.device('cpu')
        # the common shift: the first rank that has seen data (ranks with no samples contribute zeros)
        has = torch.tensor([1.0 if n else 0.0], dtype=torch.float64, device=dev)
        flags = [torch.zeros_like(has) for _ in range(dist.get_world_size(group))]
        dist.all_gather(flags, has, group=group)
        src = next((i for i, f in enumerate(flags) if float(f) != 0.0), 0)
        k = torch.from_numpy(k_r.copy()).to(dev)
        dist.broadcast(k, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        k = k.cpu().numpy()
        d = k_r - k
        s2 = s2 + np.outer(d, s1) + np.outer(s1, d) + n * np.outer(d, d)
        s1 = s1 + n * d
        buf = torch.from_numpy(np.concatenate([[float(n)], s1, s2.reshape(-1)])).to(dev)
        dist.all_reduce(buf, group=group)
        buf = buf.cpu().numpy()
        self._reduced = (int(round(buf[0])), k, buf[1:1 + self.dims], buf[1 + self.dims:].reshape(self.dims, self.dims))
        return self

    def finalize(self):
        n, shift, s1, s2 = self._host_sums()
        mu = shift + s1 / n
        sigma = (s2 - np.outer(s1, s1) / n) / (n - 1)
        return mu, sigma


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """fid_score.py:182-236: d^2 = |mu1 - mu2|^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)); the matrix square root stays on the host
    (scipy.linalg.sqrtm in float64), exactly where the reference computes it."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape, 'Training and test mean vectors have different lengths'
    assert sigma1.shape == sigma2.shape, 'Training and test covariances have different dimensions'
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        print('fid calculation produces singular product; adding %s to diagonal of cov estimates' % eps)
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError('Imaginary component {}'.format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def get_activations(images, model, batch_size=50, dims=2048, device='cuda', stats=None):
    """fid_score.py:100-180 over an in-memory uint8 / float image source: `images` is an iterable of [n, 3, H, W] batches in
    [0, 1] (device or host).  Returns the FeatureStats (the [N, dims] activation matrix is never materialised on the host)."""
    stats = stats if stats is not None else FeatureStats(dims, torch.device(device))
    for batch in images:
        pred = model(batch.to(device))[0]
        if pred.shape[2] != 1 or pred.shape[3] != 1:
            N, C, H, W = pred.shape
            rows = ops.rowsum_nc(pred)
            pred = ops.axpby(rows, 1.0 / (H * W), torch.empty_like(rows), 0.0)
        stats.update(pred.reshape(pred.shape[0], -1))
    return stats


def _image_batches(files, batch_size, device, res=None):
    """ImagePathDataset + DataLoader(shuffle=False, drop_last=False) + TF.ToTensor (fid_score.py:84-151): PIL decode on the
    host (optional Resize + CenterCrop to `res`), ToTensor on the device (dp_u8_to_float)."""
    from PIL import Image
    from . import data
    for lo in range(0, len(files), batch_size):
        arrs = []
        for f in files[lo:lo + batch_size]:
            img = Image.open(f).convert('RGB')
            if res is not None:
                img = data.resize_shorter_side(res)(img)
                w, h = img.size
                l, t = int(round((w - res) / 2.0)), int(round((h - res) / 2.0))
                img = img.crop((l, t, l + res, t + res))
            arrs.append(np.asarray(img, dtype=np.uint8))
        yield data.to_device_batch(np.stack(arrs), True, torch.device(device), data.RAW, 0.0)


def compute_statistics_of_path(path, model, batch_size, dims, device, num_samples=None, res=None):
    """fid_score.py:264-282: an .npz of (mu, sigma) or a directory of images."""
    if str(path).endswith('.npz'):
        with np.load(path) as f:
            return f['mu'][:], f['sigma'][:]
    path = pathlib.Path(path)
    files = sorted([file for ext in IMAGE_EXTENSIONS for file in path.glob('**/*.{}'.format(ext))])
    if num_samples is not None:
        files = files[:num_samples]
    print('Found %d files.' % len(files))
    return get_activations(_image_batches(files, batch_size, device, res), model, batch_size, dims, device).finalize()


def sample_to_dir(pipeline, output_dir, total_samples, batch_size, seed=0, rank=None, world=None, num_inference_steps=100,
                  stats=None, inception=None, save=True, **pipe_kwargs):
    """ddpm_sample.py:55-74: every process samples `total_samples // (batch_size * world)` batches into its own folder
    `output_dir/process_{rank}` from a generator seeded `seed + rank`, files `{i * batch_size + j}.png`.
    rank / world default to the process group's (1 process: rank 0 of 1).  With `stats` (a FeatureStats) and `inception` the
    FID features of every batch are accumulated on the device as it is produced (no PNG round trip): follow with
    `stats.all_reduce()` and the statistics cover all ranks' samples.  Returns the number of images this rank produced."""
    import torch.distributed as dist
    if rank is None or world is None:
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    sub = os.path.join(output_dir, 'process_{}'.format(rank))
    if save:
        os.makedirs(sub, exist_ok=True)
    generator = torch.Generator().manual_seed(seed + rank)
    num_batches = total_samples // (batch_size * world)
    for i in range(num_batches):
        arr = pipeline(batch_size=batch_size, num_inference_steps=num_inference_steps, generator=generator,
                       output_type='numpy', **pipe_kwargs).images                  # [B, H, W, C] in [0, 1]
        if save:
            for j, image in enumerate(pipeline.numpy_to_pil(arr)):
                image.save(os.path.join(sub, '{}.png'.format(i * batch_size + j)))
        if stats is not None:
            # what the FID reader would see: the 8-bit PNG values (numpy_to_pil rounds x * 255), as ToTensor hands them on
            u8 = torch.from_numpy((arr * 255).round().astype('uint8')).permute(0, 3, 1, 2).contiguous()
            get_activations([u8.to(torch.float32) / 255.0], inception, batch_size, stats.dims, stats.s1.device, stats=stats)
    return num_batches * batch_size


def calculate_fid_given_paths(paths, batch_size, device, dims, num_samples=None, res=None, model=None):
    """fid_score.py:285-301."""
    for p in paths:
        if not os.path.exists(p):
            raise RuntimeError('Invalid path: %s' % p)
    model = model if model is not None else InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]]).to(device)
    m1, s1 = compute_statistics_of_path(paths[0], model, batch_size, dims, device, num_samples, res)
    m2, s2 = compute_statistics_of_path(paths[1], model, batch_size, dims, device, num_samples, res)
    return calculate_frechet_distance(m1, s1, m2, s2)


def save_fid_stats(paths, batch_size, device, dims, num_samples=None, res=None, model=None):
    """fid_score.py:304-321."""
    if not os.path.exists(paths[0]):
        raise RuntimeError('Invalid path: %s' % paths[0])
    if os.path.exists(paths[1]):
        raise RuntimeError('Existing output file: %s' % paths[1])
    model = model if model is not None else InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]]).to(device)
    m1, s1 = compute_statistics_of_path(paths[0], model, batch_size, dims, device, num_samples, res)
    np.savez_compressed(paths[1], mu=m1, sigma=s1)


# ---- SSIM / MSE (ddpm_exp/compute_ssim.py) ----------------------------------------------------------------------------------
def ssim(x, y, data_range=1.0):
    """pytorch_msssim.ssim(x, y, data_range=data_range, size_average=False) -> [N] (compute_ssim.py:43)."""
    if x.device.type != 'cuda':
        raise RuntimeError('ssim runs on the MI355X HIP kernels only')
    x, y = x.to(torch.float32).contiguous(), y.to(torch.float32).contiguous()
    assert x.shape == y.shape and x.dim() == 4
    N, C, H, W = x.shape
    lib = L.load()
    part = torch.empty(max(int(lib.dp_ssim_workspace(N, C, H, W)), 1), dtype=torch.float32, device=x.device)
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    L.check(lib.dp_ssim(ops._p(x), ops._p(y), N, C, H, W, float(data_range), ops._p(part), ops._p(out), ops._stream()), 'dp_ssim')
    return out


def mse_per_image(x, y):
    """F.mse_loss(x, y, reduction='none').mean(dim=(1, 2, 3)) (compute_ssim.py:45)."""
    x, y = x.to(torch.float32).contiguous(), y.to(torch.float32).contiguous()
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    L.check(L.load().dp_mse_per_image(ops._p(x), ops._p(y), x.shape[0], x[0].numel(), ops._p(out), ops._stream()),
            'dp_mse_per_image')
    return out


def compare_directories(path1, path2, device='cuda', batch_size=100, exts=('png',)):
    """compute_ssim.py:20-53: mean SSIM and mean per-image MSE over two directories of equally named / ordered samples."""
    p1, p2 = pathlib.Path(path1), pathlib.Path(path2)
    f1 = sorted(f for e in exts for f in p1.glob('**/*.{}'.format(e)))
    f2 = sorted(f for e in exts for f in p2.glob('**/*.{}'.format(e)))
    ss, ms = [], []
    for a, b in zip(_image_batches(f1, batch_size, device), _image_batches(f2, batch_size, device)):
        ss.append(ssim(a, b, 1.0))
        ms.append(mse_per_image(a, b))
    s, m = torch.cat(ss), torch.cat(ms)
    return float(s.mean()), float(m.mean())

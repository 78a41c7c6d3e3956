"""ORACLE (test infrastructure only) -- CPU restatement of the FID / SSIM evaluation path.

Reference lines followed (relative to /root/reference):
  fid_score.py:182-236   calculate_frechet_distance   PINNED: tests/golden/fid.json holds values computed by fid_score.py itself
  fid_score.py:239-262   activation statistics (np.mean, np.cov(rowvar=False))
  inception.py:16-340    InceptionV3 wrapper + pytorch-fid's patched blocks (avg pools with count_include_pad=False,
                         max pool in the second InceptionE)
  ddpm_exp/compute_ssim.py:43-45   pytorch_msssim.ssim(data_range=1, size_average=False), per-image MSE
Third-party code the reference calls and that is absent from /root/reference and from this environment:
  * torchvision (requirements.txt, un-pinned): models.inception.Inception3 / InceptionA..E / BasicConv2d.  Restated from the
    published architecture (Szegedy et al. 2015, "Rethinking the Inception Architecture"; torchvision/models/inception.py):
    BasicConv2d = Conv2d(bias=False) -> BatchNorm2d(eps=0.001) -> ReLU.  PARITY UNPINNED for the network.
  * pytorch_msssim (un-pinned): ssim = Wang et al. 2004 with an 11-tap Gaussian (sigma 1.5), K1 = 0.01, K2 = 0.03, 'valid'
    filtering.  PARITY UNPINNED.
Never imported by the product path.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _bconv(sd, name, x, stride=1, padding=0):
    y = F.conv2d(x, sd[name + '.conv.weight'], None, stride=stride, padding=padding)
    y = F.batch_norm(y, sd[name + '.bn.running_mean'], sd[name + '.bn.running_var'], sd[name + '.bn.weight'], sd[name + '.bn.bias'],
                     False, 0.0, 0.001)
    return F.relu(y)


def _avg(x):
    return F.avg_pool2d(x, kernel_size=3, stride=1, padding=1, count_include_pad=False)


def _a(sd, n, x):
    b1 = _bconv(sd, n + '.branch1x1', x)
    b5 = _bconv(sd, n + '.branch5x5_2', _bconv(sd, n + '.branch5x5_1', x), padding=2)
    b3 = _bconv(sd, n + '.branch3x3dbl_1', x)
    b3 = _bconv(sd, n + '.branch3x3dbl_3', _bconv(sd, n + '.branch3x3dbl_2', b3, padding=1), padding=1)
    return torch.cat([b1, b5, b3, _bconv(sd, n + '.branch_pool', _avg(x))], 1)


def _b(sd, n, x):
    b3 = _bconv(sd, n + '.branch3x3', x, stride=2)
    bd = _bconv(sd, n + '.branch3x3dbl_2', _bconv(sd, n + '.branch3x3dbl_1', x), padding=1)
    bd = _bconv(sd, n + '.branch3x3dbl_3', bd, stride=2)
    return torch.cat([b3, bd, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def _c(sd, n, x):
    b1 = _bconv(sd, n + '.branch1x1', x)
    b7 = _bconv(sd, n + '.branch7x7_1', x)
    b7 = _bconv(sd, n + '.branch7x7_2', b7, padding=(0, 3))
    b7 = _bconv(sd, n + '.branch7x7_3', b7, padding=(3, 0))
    bd = _bconv(sd, n + '.branch7x7dbl_1', x)
    bd = _bconv(sd, n + '.branch7x7dbl_2', bd, padding=(3, 0))
    bd = _bconv(sd, n + '.branch7x7dbl_3', bd, padding=(0, 3))
    bd = _bconv(sd, n + '.branch7x7dbl_4', bd, padding=(3, 0))
    bd = _bconv(sd, n + '.branch7x7dbl_5', bd, padding=(0, 3))
    return torch.cat([b1, b7, bd, _bconv(sd, n + '.branch_pool', _avg(x))], 1)


def _d(sd, n, x):
    b3 = _bconv(sd, n + '.branch3x3_2', _bconv(sd, n + '.branch3x3_1', x), stride=2)
    b7 = _bconv(sd, n + '.branch7x7x3_1', x)
    b7 = _bconv(sd, n + '.branch7x7x3_2', b7, padding=(0, 3))
    b7 = _bconv(sd, n + '.branch7x7x3_3', b7, padding=(3, 0))
    b7 = _bconv(sd, n + '.branch7x7x3_4', b7, stride=2)
    return torch.cat([b3, b7, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def _e(sd, n, x, pool):
    b1 = _bconv(sd, n + '.branch1x1', x)
    t = _bconv(sd, n + '.branch3x3_1', x)
    b3 = torch.cat([_bconv(sd, n + '.branch3x3_2a', t, padding=(0, 1)), _bconv(sd, n + '.branch3x3_2b', t, padding=(1, 0))], 1)
    t = _bconv(sd, n + '.branch3x3dbl_2', _bconv(sd, n + '.branch3x3dbl_1', x), padding=1)
    bd = torch.cat([_bconv(sd, n + '.branch3x3dbl_3a', t, padding=(0, 1)), _bconv(sd, n + '.branch3x3dbl_3b', t, padding=(1, 0))], 1)
    p = _avg(x) if pool == 'avg' else F.max_pool2d(x, kernel_size=3, stride=1, padding=1)      # inception.py:296-300 / 328-333
    return torch.cat([b1, b3, bd, _bconv(sd, n + '.branch_pool', p)], 1)


@torch.no_grad()
def inception_forward(sd, inp, output_blocks=(3,), resize_input=True, normalize_input=True):
    """inception.py:129-163.  sd: state dict with torchvision's Inception3 key names."""
    x = inp
    if resize_input:
        x = F.interpolate(x, size=(299, 299), mode='bilinear', align_corners=False)
    if normalize_input:
        x = 2 * x - 1
    out = []
    last = max(output_blocks)
    x = _bconv(sd, 'Conv2d_1a_3x3', x, stride=2)
    x = _bconv(sd, 'Conv2d_2a_3x3', x)
    x = _bconv(sd, 'Conv2d_2b_3x3', x, padding=1)
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    if 0 in output_blocks:
        out.append(x)
    if last >= 1:
        x = _bconv(sd, 'Conv2d_4a_3x3', _bconv(sd, 'Conv2d_3b_1x1', x))
        x = F.max_pool2d(x, kernel_size=3, stride=2)
        if 1 in output_blocks:
            out.append(x)
    if last >= 2:
        for n in ('Mixed_5b', 'Mixed_5c', 'Mixed_5d'):
            x = _a(sd, n, x)
        x = _b(sd, 'Mixed_6a', x)
        for n in ('Mixed_6b', 'Mixed_6c', 'Mixed_6d', 'Mixed_6e'):
            x = _c(sd, n, x)
        if 2 in output_blocks:
            out.append(x)
    if last >= 3:
        x = _d(sd, 'Mixed_7a', x)
        x = _e(sd, 'Mixed_7b', x, 'avg')
        x = _e(sd, 'Mixed_7c', x, 'max')
        out.append(F.adaptive_avg_pool2d(x, (1, 1)))
    return out


def activation_statistics(act):
    """fid_score.py:258-261."""
    act = np.asarray(act, dtype=np.float64)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """fid_score.py:182-236."""
    from scipy import linalg
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def ssim(x, y, data_range=1.0, win_size=11, sigma=1.5, K=(0.01, 0.03)):
    """pytorch_msssim.ssim(x, y, data_range, size_average=False) restated (float64 internally): per-image SSIM."""
    x, y = x.double(), y.double()
    coords = torch.arange(win_size, dtype=torch.float64) - win_size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    g = (g / g.sum())
    C = x.shape[1]

    def gf(t):
        t = F.conv2d(t, g.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
        return F.conv2d(t, g.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = gf(x), gf(y)
    s1, s2, s12 = gf(x * x) - mu1 * mu1, gf(y * y) - mu2 * mu2, gf(x * y) - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    m = ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs
    return m.flatten(2).mean(-1).mean(1)

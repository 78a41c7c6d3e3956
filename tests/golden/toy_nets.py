"""Small plain-PyTorch networks for the generic tracer (diff-pruning_amd/trace.py).

Written for this repository (nothing here comes from the reference).  `make_golden.py do_traced` runs the REFERENCE's
DependencyGraph / MagnitudePruner on them and records group tables, pruning histories and pruned shapes in
traced_groups.json; tests/test_cpu.py runs this package's tracer on the same networks and compares.
Each entry of NETS: name -> (constructor, example-input builder, names of the layers to ignore (the output heads))."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PlainCNN(nn.Module):
    """conv-bn-relu x2 -> maxpool -> flatten -> linear -> relu -> linear: BatchNorm members and the flatten index map."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 8, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(8)
        self.conv2 = nn.Conv2d(8, 12, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(12)
        self.fc1 = nn.Linear(12 * 4 * 4, 20)
        self.fc2 = nn.Linear(20, 5)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(F.relu(self.bn2(self.conv2(x))), 2)
        x = torch.flatten(x, 1)
        return self.fc2(F.relu(self.fc1(x)))


class ResCat(nn.Module):
    """GroupNorm residual block, a skip concatenation, and a split whose parts feed different convolutions."""

    def __init__(self, swap=False):
        super().__init__()
        self.swap = swap
        self.conv_in = nn.Conv2d(3, 16, 3, padding=1)
        self.norm1 = nn.GroupNorm(4, 16)
        self.conv1 = nn.Conv2d(16, 16, 3, padding=1)
        self.norm2 = nn.GroupNorm(4, 16)
        self.conv2 = nn.Conv2d(16, 16, 3, padding=1)
        self.down = nn.Conv2d(16, 24, 3, stride=2, padding=1)
        self.mid = nn.Conv2d(24, 24, 3, padding=1)
        self.up = nn.Conv2d(24, 8, 3, padding=1)
        self.fuse = nn.Conv2d(8 + 16, 20, 1)
        self.left = nn.Conv2d(10, 6, 3, padding=1)
        self.right = nn.Conv2d(10, 6, 1)
        self.head = nn.Conv2d(6, 3, 1)

    def forward(self, x):
        x = self.conv_in(x)
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        skip = x + h
        d = self.mid(F.relu(self.down(skip)))
        u = self.up(F.interpolate(d, scale_factor=2.0, mode='nearest'))
        f = self.fuse(torch.cat([u, skip], dim=1))
        a, b = f.split([self.left.in_channels, self.right.in_channels], dim=1)     # sizes follow the pruned layers
        if self.swap:       # the reference numbers the outputs of a split in TRACE order (dependency.py:825-853), which is
            return self.head(self.left(a) + self.right(b))      # the reverse of the chunk order here: see test_cpu
        return self.head(self.right(b) + self.left(a))


class TokenMixer(nn.Module):
    """[B, T, C] tokens: LayerNorm, single-head attention through F.scaled_dot_product_attention and a GEGLU feed-forward
    (chunk on the LAST dim).  token_cat=True adds a token-wise concatenation (a class token) that must not be mistaken
    for a channel concatenation: the reference's tracer does mistake it (dependency.py:690-705 special-cases only the
    unwrapped-parameter form), so that variant is checked against hand-derived groups instead of a reference fixture."""

    def __init__(self, token_cat=False):
        super().__init__()
        self.token_cat = token_cat
        self.embed = nn.Linear(6, 16)
        self.cls = nn.Linear(6, 16) if token_cat else None
        self.norm1 = nn.LayerNorm(16)
        self.to_q = nn.Linear(16, 16, bias=False)
        self.to_k = nn.Linear(16, 16, bias=False)
        self.to_v = nn.Linear(16, 16, bias=False)
        self.to_out = nn.Linear(16, 16)
        self.norm2 = nn.LayerNorm(16)
        self.proj = nn.Linear(16, 2 * 24)
        self.ff_out = nn.Linear(24, 16)
        self.head = nn.Linear(16, 4)

    def forward(self, x):
        h = self.embed(x)
        if self.token_cat:
            h = torch.cat([self.cls(x[:, :1]), h], dim=1)                  # one more token, same channels
        n = self.norm1(h)
        h = h + self.to_out(F.scaled_dot_product_attention(self.to_q(n), self.to_k(n), self.to_v(n)))
        a, g = self.proj(self.norm2(h)).chunk(2, dim=-1)
        h = h + self.ff_out(a * F.gelu(g))
        return self.head(h.mean(dim=1))


class MobileBlock(nn.Module):
    """pointwise -> depthwise -> pointwise with BatchNorm and a residual: depthwise convolutions as pass-through members."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Conv2d(3, 8, 3, padding=1)
        self.expand = nn.Conv2d(8, 24, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(24)
        self.dw = nn.Conv2d(24, 24, 3, padding=1, groups=24, bias=False)
        self.bn2 = nn.BatchNorm2d(24)
        self.project = nn.Conv2d(24, 8, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(8)
        self.head = nn.Conv2d(8, 2, 1)

    def forward(self, x):
        x = self.stem(x)
        h = F.relu6(self.bn1(self.expand(x)))
        h = F.relu6(self.bn2(self.dw(h)))
        return self.head(x + self.bn3(self.project(h)))


class TimeCondUNet(nn.Module):
    """A two-level UNet with a time-embedding MLP added to the features, a generator-style Linear -> view(N, C, H, W)
    bias map (the inverse flatten), an input-image concatenation and two outputs.  image_first=True puts the (gradient-
    free) image FIRST in the concatenation: the reference's trace never sees that input and computes the offsets of the
    inputs after it 3 channels too low, so that variant is checked semantically (pruned == zeroed) instead."""

    def __init__(self, image_first=False):
        super().__init__()
        self.image_first = image_first
        self.t1 = nn.Linear(8, 16)
        self.t2 = nn.Linear(16, 16)
        self.conv_in = nn.Conv2d(3, 12, 3, padding=1)
        self.norm = nn.GroupNorm(3, 12)
        self.conv_a = nn.Conv2d(12, 12, 3, padding=1)
        self.temb = nn.Linear(16, 12)
        self.down = nn.Conv2d(12, 18, 3, stride=2, padding=1)
        self.seed = nn.Linear(16, 18 * 4 * 4)
        self.mid = nn.Conv2d(18, 18, 3, padding=1)
        self.up = nn.Conv2d(18, 12, 3, padding=1)
        self.merge = nn.Conv2d(3 + 12 + 12, 10, 3, padding=1)
        self.out_a = nn.Conv2d(10, 3, 1)
        self.out_b = nn.Conv2d(10, 2, 1)      # (a 1-channel conv has groups == out_channels: the reference types it depthwise)

    def forward(self, x, t):
        e = self.t2(F.silu(self.t1(t)))
        h0 = self.conv_in(x)
        h = self.conv_a(F.silu(self.norm(h0))) + self.temb(F.silu(e))[:, :, None, None]
        d = self.down(h) + self.seed(e).view(e.shape[0], -1, 4, 4)
        u = self.up(F.interpolate(self.mid(F.silu(d)), scale_factor=2.0, mode='nearest'))
        parts = [x, u, h0] if self.image_first else [u, h0, x]
        m = F.silu(self.merge(torch.cat(parts, dim=1)))
        return self.out_a(m), self.out_b(m)


class Generator(nn.Module):
    """Class-conditional generator: Embedding -> Linear -> view(N, C, 4, 4) -> ConvTranspose2d -> InstanceNorm2d(affine) ->
    PReLU(per channel) -> Conv2d -> PReLU(one shared slope) -> Conv2d: the four further layer kinds of the tracer."""

    def __init__(self):
        super().__init__()
        self.embed = nn.Embedding(10, 12)
        self.fc = nn.Linear(12, 8 * 4 * 4)
        self.up = nn.ConvTranspose2d(8, 10, 4, stride=2, padding=1)
        self.norm = nn.InstanceNorm2d(10, affine=True)
        self.act = nn.PReLU(10)
        self.conv = nn.Conv2d(10, 6, 3, padding=1)
        self.act2 = nn.PReLU()
        self.head = nn.Conv2d(6, 3, 1)

    def forward(self, labels):
        h = self.fc(self.embed(labels))
        h = self.up(h.view(h.shape[0], -1, 4, 4))
        h = self.act(self.norm(h))
        return self.head(self.act2(self.conv(h)))


class ZeroEquiv(nn.Module):
    """ReLU-only, norm-free network (so that removing a channel == zeroing the layer that produces it) with an image-first
    concatenation, a three-way uneven split, a flatten and a depthwise convolution: the semantic check of the tracer."""

    def __init__(self):
        super().__init__()
        self.conv_in = nn.Conv2d(3, 8, 3, padding=1)
        self.conv_a = nn.Conv2d(8, 6, 3, padding=1)
        self.merge = nn.Conv2d(3 + 6 + 8, 12, 3, padding=1)
        self.p = nn.Conv2d(3, 5, 1)
        self.q = nn.Conv2d(5, 5, 3, padding=1)
        self.dw = nn.Conv2d(4, 4, 3, padding=1, groups=4)
        self.r = nn.Conv2d(4, 5, 1)
        self.fc = nn.Linear(5 * 4 * 4, 7)
        self.head = nn.Linear(7, 3)

    def forward(self, x):
        h0 = F.relu(self.conv_in(x))
        a = F.relu(self.conv_a(h0))
        m = F.relu(self.merge(torch.cat([x, a, h0], dim=1)))
        s0, s1, s2 = m.split([self.p.in_channels, self.q.in_channels, self.dw.in_channels], dim=1)
        y = F.relu(self.p(s0)) + F.relu(self.q(s1)) + F.relu(self.r(F.relu(self.dw(s2))))
        y = F.max_pool2d(y, 2)
        return self.head(F.relu(self.fc(torch.flatten(y, 1))))


def _img(c=3, h=8):
    return lambda: (torch.linspace(-1, 1, 2 * c * h * h).reshape(2, c, h, h),)


NETS = {
    'plain_cnn': (PlainCNN, _img(3, 8), ['fc2']),
    'res_cat': (ResCat, _img(3, 8), ['head']),
    'token_mixer': (TokenMixer, lambda: (torch.linspace(-1, 1, 2 * 5 * 6).reshape(2, 5, 6),), ['head']),
    'mobile_block': (MobileBlock, _img(3, 8), ['head']),
    'generator': (Generator, lambda: (torch.tensor([1, 7]),), ['head']),
    'time_cond_unet': (TimeCondUNet, lambda: (torch.linspace(-1, 1, 2 * 3 * 64).reshape(2, 3, 8, 8),
                                              torch.linspace(0, 1, 16).reshape(2, 8)), ['out_a', 'out_b']),
}


def build(name, seed=0, **kw):
    ctor, inputs, ignored = NETS[name]
    torch.manual_seed(seed)
    model = ctor(**kw).eval()
    return model, inputs(), [dict(model.named_modules())[n] for n in ignored]


class IndexScore:
    """A deterministic stand-in importance (no gradients, no device kernels): score of channel i of an n-channel group is
    ((i * 37 + 11 * n) % 101) -- the same callable drives the reference's pruner and this package's."""

    def __call__(self, group, ch_groups=1):
        n = len(group[0][1])
        return torch.tensor([float((i * 37 + 11 * n) % 101) for i in range(n)])

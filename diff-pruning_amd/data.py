"""Data input pipeline: the datasets and transforms of the reference's training scripts with the
per-pixel work on the device.

  utils.py:8-24      UnlabeledImageFolder           -> UnlabeledImageFolder (recursive glob per extension; see the class for the
                     one deliberate deviation from the reference's pattern)
  utils.py:31-58     get_dataset: CIFAR-10 = RandomHorizontalFlip + ToTensor + Normalize(0.5, 0.5);
                     image folders = Resize(256) + RandomCrop(256) + the same three
  ddpm_exp/datasets/__init__.py:30-60,176-192   Resize + flip + ToTensor, then data_transform (uniform dequantization, 2x - 1)
  ddpm_exp/datasets/__init__.py:60-152, celeba.py:50-139   config-driven datasets -> dataset_from_config (CIFAR10, CELEBA: split
                     file, 128 x 128 crop window, resize); LSUN / FFHQ (datasets/lsun.py, ffhq.py: LMDB environments) -> Lsun / FfhqLmdb over
                     lmdb_reader.Environment, a pure-Python reader of the data.mdb format
  utils.py:41-49     CIFAR-100              -> Cifar100Batches
  ddpm_train.py:313-318   DataLoader(shuffle=True)  -> DeviceLoader

Host side (decode, resize, crop, shuffle) stays on the CPU like torchvision's PIL transforms; the device receives uint8 pixels
(a quarter of the fp32 bytes over PCIe) and ONE HIP kernel (dp_u8_to_float) does ToTensor + flip + normalisation.  Random
decisions (flip per image, dequantisation noise, crop offsets, shuffling) are counter-based -- functions of (seed, epoch, global
sample index) -- so every rank of a data-parallel run can build its own shard of the same global batch without communication.
torchvision is not part of this repository's environment: the three transforms are restated from their definitions
(ToTensor = x / 255; Normalize = (x - mean) / std; RandomHorizontalFlip = reverse W with probability p).
"""
import glob
import math
import os
import pickle

import numpy as np
import torch

from . import ops, _lib as L

NORMALIZE, RESCALE, RAW = 1, 2, 0


class UnlabeledImageFolder:
    """utils.py:8-24.  `transform` receives and returns a PIL image (host-side Resize / crop); items are uint8 HWC arrays."""

    def __init__(self, root, transform=None, exts=("*.jpg", "*.png", "*.jpeg", "*.webp")):
        self.root, self.transform = root, transform
        self.files = []
        # The reference formats '**/*.{}'.format("*.jpg") -> '**/*.*.jpg' (utils.py:16), which only matches names with TWO
        # dots ("a.b.jpg"): on an ordinary folder it finds nothing and training silently iterates zero batches.  Deliberate
        # deviation: '**/' + ext finds every file with the extension (a superset of the reference's matches), in sorted order
        # per extension so that every rank of a data-parallel run indexes the same list; an empty folder is an error.
        for ext in exts:
            self.files.extend(sorted(glob.glob(os.path.join(root, '**', ext), recursive=True)))
        if not self.files:
            raise FileNotFoundError('no image files (%s) under %r' % (', '.join(exts), root))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        from PIL import Image
        img = Image.open(self.files[idx]).convert('RGB')
        if self.transform is not None:
            img = self.transform(img)
        return np.asarray(img, dtype=np.uint8)


class Cifar10Batches:
    """The python pickles torchvision's CIFAR10 reads (`cifar-10-batches-py/data_batch_1..5`, `test_batch`): uint8 [N, 3, 32, 32]."""
    hwc = False

    def __init__(self, root, train=True):
        base = os.path.join(root, 'cifar-10-batches-py')
        names = ['data_batch_%d' % i for i in range(1, 6)] if train else ['test_batch']
        data = []
        for n in names:
            with open(os.path.join(base, n), 'rb') as f:
                entry = pickle.load(f, encoding='latin1')
            data.append(np.asarray(entry['data'], dtype=np.uint8))
        self.data = np.concatenate(data).reshape(-1, 3, 32, 32)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx]


class Cifar100Batches(Cifar10Batches):
    """utils.py:41-49 (`torchvision.datasets.CIFAR100`): `cifar-100-python/train` / `test`, same record layout as CIFAR-10."""

    def __init__(self, root, train=True):
        with open(os.path.join(root, 'cifar-100-python', 'train' if train else 'test'), 'rb') as f:
            entry = pickle.load(f, encoding='latin1')
        self.data = np.asarray(entry['data'], dtype=np.uint8).reshape(-1, 3, 32, 32)


CELEBA_WINDOW = (25, 57, 153, 185)      # PIL box (left, upper, right, lower) of Crop(x1=57, x2=185, y1=25, y2=153)


class CelebAAligned:
    """ddpm_exp/datasets/celeba.py:50-107,137-139 with the transform of ddpm_exp/datasets/__init__.py:60-93: the aligned
    178 x 218 JPEGs `root/Img/img_align_celeba/<name>` of one split of `root/Eval/list_eval_partition.txt` (`<name> <0|1|2>`
    per line: train / valid / test), each cut to the 128 x 128 window around (cx, cy) = (89, 121) -- `Crop(x1, x2, y1, y2)`
    calls `F.crop(img, top=x1, left=y1, height=x2 - x1, width=y2 - y1)`, i.e. rows 57..184 and columns 25..152 -- and
    resized to `image_size` (bilinear, as transforms.Resize on a PIL image).  Attributes / identities / landmarks, which the
    class also parses, are not read: the diffusion runners discard the target (`for i, (x, y) in enumerate(train_loader)`)."""
    _SPLITS = {'train': 0, 'valid': 1, 'test': 2}

    def __init__(self, root, split='train', image_size=64):
        if split.lower() not in self._SPLITS:
            raise ValueError('Wrong split entered! Please use split="train" or split="valid" or split="test"')
        want = self._SPLITS[split.lower()]
        self.root, self.image_size = root, int(image_size)
        self.files = []
        with open(os.path.join(root, 'Eval', 'list_eval_partition.txt')) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 2 and int(parts[1]) == want:
                    self.files.append(parts[0])
        if not self.files:
            raise FileNotFoundError('split %r of %s lists no image' % (split, os.path.join(root, 'Eval', 'list_eval_partition.txt')))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        from PIL import Image
        img = Image.open(os.path.join(self.root, 'Img', 'img_align_celeba', self.files[idx])).convert('RGB')
        img = img.crop(CELEBA_WINDOW)
        if img.size != (self.image_size, self.image_size):
            img = img.resize((self.image_size, self.image_size), Image.BILINEAR)
        return np.asarray(img, dtype=np.uint8)


def center_crop(size):
    """transforms.CenterCrop(size) on a PIL image (the image is at least `size` on both sides after Resize(size)): torchvision's
    rounding, top = round((h - size) / 2), left = round((w - size) / 2)."""
    def f(img):
        w, h = img.size
        left, top = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
        return img.crop((left, top, left + size, top + size))
    return f


LSUN_CATEGORIES = ('bedroom', 'bridge', 'church_outdoor', 'classroom', 'conference_room', 'dining_room', 'kitchen', 'living_room',
                   'restaurant', 'tower')


class LsunClassLmdb:
    """One `<category>_<split>_lmdb` environment of LSUN (ddpm_exp/datasets/lsun.py:11-52): values are encoded images, the item
    order is the key order of the environment -- cached, as the reference caches it, in the pickle `_cache_<dir name>` next to the
    directory.  Items are uint8 HWC arrays after `transform` (PIL -> PIL).  Read through lmdb_reader.Environment."""

    def __init__(self, root, transform=None):
        from .lmdb_reader import Environment
        self.root, self.transform = root, transform
        self.env = Environment(root)
        self.length = self.env.stat()['entries']
        parts = root.rstrip('/').split('/')
        cache_file = os.path.join('/'.join(parts[:-1]), '_cache_' + parts[-1])
        if os.path.isfile(cache_file):
            with open(cache_file, 'rb') as f:
                self.keys = pickle.load(f)
        else:
            self.keys = self.env.keys()
            try:
                with open(cache_file, 'wb') as f:
                    pickle.dump(self.keys, f)
            except OSError:
                pass                                    # a read-only dataset directory: the key walk is repeated next time

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        import io
        from PIL import Image
        buf = self.env.get(self.keys[idx])
        if buf is None:
            raise KeyError(self.keys[idx])
        img = Image.open(io.BytesIO(buf)).convert('RGB')
        if self.transform is not None:
            img = self.transform(img)
        return np.asarray(img, dtype=np.uint8)


class Lsun:
    """ddpm_exp/datasets/lsun.py:55-173: `classes` is 'train' / 'val' (every category), 'test', or a list such as
    ['bedroom_train']; item i belongs to the first class database whose cumulative length exceeds i.  The category label the
    reference returns next to the image is discarded by the diffusion runners and not produced here."""

    def __init__(self, root, classes='train', transform=None):
        self.classes = self._verify_classes(classes)
        self.dbs = [LsunClassLmdb(root + '/' + c + '_lmdb', transform) for c in self.classes]
        self.indices, count = [], 0
        for db in self.dbs:
            count += len(db)
            self.indices.append(count)
        self.length = count

    @staticmethod
    def _verify_classes(classes):
        splits = ('train', 'val', 'test')
        if isinstance(classes, str):
            if classes not in splits:
                raise ValueError("Unknown value '%s' for argument classes. Valid values are {%s}." % (classes, ', '.join(splits)))
            return [classes] if classes == 'test' else [c + '_' + classes for c in LSUN_CATEGORIES]
        out = list(classes)
        for c in out:
            if not isinstance(c, str):
                raise ValueError('Expected type str for elements in argument classes, but got type %s.' % type(c))
            category, _, split = c.rpartition('_')
            if category not in LSUN_CATEGORIES:
                raise ValueError("Unknown value '%s' for LSUN class. Valid values are {%s}." % (category, ', '.join(LSUN_CATEGORIES)))
            if split not in splits:
                raise ValueError("Unknown value '%s' for postfix. Valid values are {%s}." % (split, ', '.join(splits)))
        return out

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        sub, which = 0, 0
        for ind in self.indices:
            if index < ind:
                break
            which += 1
            sub = ind
        return self.dbs[which][index - sub]


class FfhqLmdb:
    """ddpm_exp/datasets/ffhq.py:8-40: one environment holding every resolution; `length` under the key b'length', image i of
    resolution r under b'<r>-<i zero-filled to 5 digits>'."""

    def __init__(self, path, resolution=8, transform=None):
        from .lmdb_reader import Environment
        self.env = Environment(path)
        n = self.env.get(b'length')
        if n is None:
            raise IOError('Cannot open lmdb dataset', path)
        self.length, self.resolution, self.transform = int(n.decode('utf-8')), int(resolution), transform

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        import io
        from PIL import Image
        key = ('%d-%s' % (self.resolution, str(index).zfill(5))).encode('utf-8')
        buf = self.env.get(key)
        if buf is None:
            raise KeyError(key)
        img = Image.open(io.BytesIO(buf)).convert('RGB')
        if self.transform is not None:
            img = self.transform(img)
        return np.asarray(img, dtype=np.uint8)


def ffhq_split_indices(num_items):
    """ddpm_exp/datasets/__init__.py:166-177: the 90 / 10 split of FFHQ -- list(range(N)) shuffled by the LEGACY global numpy
    generator seeded with 2019 (the caller's generator state is saved and restored around it), first int(0.9 N) indices train,
    the rest test.  Both branches: training never sees the held-out tenth."""
    indices = list(range(num_items))
    state = np.random.get_state()
    np.random.seed(2019)
    np.random.shuffle(indices)
    np.random.set_state(state)
    cut = int(num_items * 0.9)
    return indices[:cut], indices[cut:]


class IndexSubset:
    """torch.utils.data.Subset for the array-returning datasets of this module."""

    def __init__(self, dataset, indices):
        self.dataset, self.indices = dataset, list(indices)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        return self.dataset[self.indices[i]]


class ArrayDataset:
    """uint8 images already in memory: [N, H, W, C] (hwc=True) or [N, C, H, W]."""

    def __init__(self, array, hwc=True):
        self.data, self.hwc = np.ascontiguousarray(array, dtype=np.uint8), hwc

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx]


def resize_shorter_side(size):
    """transforms.Resize(size) on a PIL image: shorter side -> size, bilinear (torchvision's default for PIL inputs)."""
    def f(img):
        from PIL import Image
        w, h = img.size
        if (w <= h and w == size) or (h <= w and h == size):
            return img
        if w < h:
            return img.resize((size, int(size * h / w)), Image.BILINEAR)
        return img.resize((int(size * w / h), size), Image.BILINEAR)
    return f


def epoch_permutation(n, seed, epoch):
    """DataLoader(shuffle=True): a permutation of the dataset per epoch, identical on every rank."""
    return np.random.default_rng([int(seed) & 0xFFFFFFFF, int(epoch)]).permutation(n)


def crop_offsets(indices, heights, widths, crop, seed, epoch):
    """RandomCrop(crop) offsets, drawn PER IMAGE from that image's own size as torchvision does (`RandomCrop.get_params`:
    i in [0, h - crop], j in [0, w - crop]) -- after Resize(256) on the shorter side a folder holds 256 x 341 and 341 x 256
    images side by side.  Counter-based: sample `g` (global index within the epoch) always gets the same pair, whichever rank
    or batch asks, and a batch costs O(batch) draws."""
    ys, xs = np.zeros(len(indices), dtype=np.int64), np.zeros(len(indices), dtype=np.int64)
    for k, (g, h, w) in enumerate(zip(indices, heights, widths)):
        if h < crop or w < crop:
            raise ValueError('RandomCrop(%d) on a %d x %d image' % (crop, h, w))
        u = np.random.default_rng([int(seed) & 0xFFFFFFFF, int(epoch), 0xC0, int(g)]).random(2)
        ys[k], xs[k] = int(u[0] * (h - crop + 1)), int(u[1] * (w - crop + 1))
    return ys, xs


def to_device_batch(u8, hwc, device, mode=NORMALIZE, flip_p=0.5, seed=0, epoch=0, n_off=0, dequant=False, out=None):
    """uint8 batch (numpy or torch, host or device) -> fp32 [N, C, H, W] on `device` through dp_u8_to_float.
    n_off: global index of the first sample (flip decisions and noise are functions of the global index)."""
    if device.type != 'cuda':
        raise RuntimeError('the input pipeline kernel runs on the MI355X only')
    t = torch.as_tensor(u8)
    if t.dtype != torch.uint8 or t.dim() != 4:
        raise ValueError('expected a 4-D uint8 batch')
    if t.device.type != 'cuda':
        t = t.contiguous().pin_memory().to(device, non_blocking=True)
    t = t.contiguous()
    if hwc:
        N, H, W, Cc = t.shape
    else:
        N, Cc, H, W = t.shape
    if out is None:
        out = ops.empty_act((N, Cc, H, W), device)
    d = L.Dropout()
    d.seed, d.site, d.step, d.n_off = int(seed) & 0xFFFFFFFFFFFFFFFF, 0xF11B, int(epoch) & 0xFFFFFFFF, int(n_off)
    thr = int(math.ceil(flip_p * (1 << 24))) if flip_p else 0
    import ctypes as C
    L.check(L.load().dp_u8_to_float(C.c_void_p(t.data_ptr()), 1 if hwc else 0, N, Cc, H, W, C.c_void_p(out.data_ptr()),
                                    out.stride(0), mode, thr, 1 if dequant else 0, C.byref(d), ops._stream()),
            'dp_u8_to_float')
    out._keepalive = t            # the uint8 staging buffer must outlive the asynchronous kernel
    return out


class DeviceLoader:
    """`for batch in DeviceLoader(dataset, 128, device)`: shuffled (optional), sharded over ranks, decoded on the host,
    transformed on the device.  rank / world: this rank takes samples [rank * B, (rank + 1) * B) of every global batch of
    world * B (ddpm_train.py's per-device train_batch_size under accelerate)."""

    def __init__(self, dataset, batch_size, device, shuffle=True, seed=0, mode=NORMALIZE, flip_p=0.5, crop=None,
                 dequant=False, rank=0, world=1, drop_last=False):
        self.ds, self.B, self.device = dataset, batch_size, torch.device(device)
        self.shuffle, self.seed, self.mode, self.flip_p, self.crop, self.dequant = shuffle, seed, mode, flip_p, crop, dequant
        self.rank, self.world, self.drop_last = rank, world, drop_last
        self.epoch = 0

    def __len__(self):
        g = self.B * self.world
        n = len(self.ds)
        return n // g if self.drop_last else -(-n // g)

    def __iter__(self):
        n = len(self.ds)
        order = epoch_permutation(n, self.seed, self.epoch) if self.shuffle else np.arange(n)
        g = self.B * self.world
        hwc = getattr(self.ds, 'hwc', True)
        for b in range(len(self)):
            lo = b * g + self.rank * self.B
            idx = order[lo:min(lo + self.B, (b + 1) * g, n)]
            if len(idx) == 0:
                return
            items = [self.ds[int(i)] for i in idx]
            if self.crop is not None:                                   # transforms.RandomCrop(crop) on HWC arrays
                ys, xs = crop_offsets(range(lo, lo + len(items)), [it.shape[0] for it in items], [it.shape[1] for it in items],
                                      self.crop, self.seed, self.epoch)
                items = [it[y:y + self.crop, x:x + self.crop] for it, y, x in zip(items, ys, xs)]
            yield to_device_batch(np.stack(items), hwc, self.device, self.mode, self.flip_p, self.seed, self.epoch, lo,
                                  self.dequant)
        self.epoch += 1


def get_dataset(name_or_path, root='./data'):
    """utils.py:31-58 (the transforms live in DeviceLoader / the device kernel): 'cifar10', 'cifar100' or an image directory."""
    if name_or_path.lower() == 'cifar10':
        return Cifar10Batches(os.path.join(root)), dict(crop=None)
    if name_or_path.lower() == 'cifar100':
        return Cifar100Batches(os.path.join(root)), dict(crop=None)
    if os.path.isdir(name_or_path):
        return UnlabeledImageFolder(name_or_path, transform=resize_shorter_side(256)), dict(crop=256)
    raise ValueError('unknown dataset %r' % (name_or_path,))



def dataset_from_config(data, root='data', train=True):
    """ddpm_exp/datasets/__init__.py:30-152 for the `data:` block of a ddpm_exp config (dict): returns (dataset, DeviceLoader
    keywords).  The host side produces uint8 images of `image_size`; flip (`random_flip`), ToTensor, `uniform_dequantization` and
    `rescaled` (2x - 1) are the device kernel's job (data_transform, :160-174).  CIFAR10 and CELEBA read the files torchvision
    reads; LSUN and FFHQ are LMDB environments read by this package's own pure-Python reader (lmdb_reader.py: `lmdb` is not part of
    this environment; format restated from LMDB 0.9, unpinned)."""
    name = str(data['dataset']).upper()
    size = int(data.get('image_size', 32))
    if data.get('gaussian_dequantization') or data.get('logit_transform'):
        raise NotImplementedError('gaussian_dequantization / logit_transform: no shipped config enables them')
    kw = dict(mode=RESCALE if data.get('rescaled', True) else RAW, flip_p=0.5 if (train and data.get('random_flip', True)) else 0.0,
              dequant=bool(data.get('uniform_dequantization', False)), crop=None)
    if name == 'CIFAR10':
        ds = Cifar10Batches(os.path.join(root, 'cifar10'), train=train)
        if size != 32:
            raise NotImplementedError('CIFAR10 at image_size %d: the shipped configs use 32' % size)
        return ds, kw
    if name == 'CELEBA':
        return CelebAAligned(os.path.join(root, 'celeba'), 'train' if train else 'test', size), kw
    if name == 'LSUN':                               # __init__.py:109-140: <category>_train / _val, Resize + CenterCrop
        def tf(img, _r=resize_shorter_side(size), _c=center_crop(size)):
            return _c(_r(img))
        return Lsun(os.path.join(root, 'lsun'), ['%s_%s' % (data['category'], 'train' if train else 'val')], tf), kw
    if name == 'FFHQ':                               # __init__.py:142-157: the stored resolution, no resize
        ds = FfhqLmdb(os.path.join(root, 'FFHQ'), size)
        train_idx, test_idx = ffhq_split_indices(len(ds))
        return IndexSubset(ds, train_idx if train else test_idx), kw
    raise ValueError('unknown dataset %r' % (data['dataset'],))


def inverse_data_transform(x, rescaled=True):
    """ddpm_exp/datasets/__init__.py:177-186 for the shipped configs (no image_mean, no logit transform): samples in [-1, 1]
    -> [0, 1], clamped.  A few elementwise torch ops on the sampler's OUTPUT (image saving), not part of the timed path."""
    if rescaled:
        x = (x + 1.0) / 2.0
    return torch.clamp(x, 0.0, 1.0)

"""Read-only reader of an LMDB environment (`data.mdb`), pure Python over mmap.

The reference reads its LSUN and FFHQ images through the `lmdb` module (ddpm_exp/datasets/lsun.py:13-52: `lmdb.open(root,
readonly=True, lock=False)`, `txn.stat()["entries"]`, `txn.cursor()` over the keys, `txn.get(key)`; ffhq.py:9-40: `txn.get(b'length')`,
`txn.get(b'<resolution>-<index zero-filled to 5>')`).  liblmdb / py-lmdb are third-party code absent from the reference tree and from
this image, so the four operations those files use are restated here from LMDB 0.9's published on-disk format (mdb.c: MDB_meta,
MDB_db, MDB_page, MDB_node) -- **format parity unpinned**: no environment written by liblmdb exists on this machine; the tests
build environments with a writer that follows the same structure definitions (tests/helpers.write_lmdb).

Layout (64-bit, little endian):
  page header (16 bytes): pgno u64 | pad u16 | flags u16 | lower u16, upper u16 (overflow pages: page count u32 instead)
  flags: 0x01 branch, 0x02 leaf, 0x04 overflow, 0x08 meta;  keys on a page = (lower - 16) / 2, u16 node offsets follow the header
  meta pages 0 and 1 (the one with the larger txnid is current), after the header:
      magic u32 = 0xBEEFC0DE | version u32 | address u64 | mapsize u64 | MDB_db free | MDB_db main | last_pg u64 | txnid u64
      MDB_db (48 bytes): pad u32 (free DB: the page size) | flags u16 | depth u16 | branch, leaf, overflow pages u64 x 3 |
                         entries u64 | root u64 (all ones: empty)
  node (8-byte header): lo u16, hi u16, flags u16, ksize u16, key bytes, data bytes
      branch node: child page = lo | hi << 16 | flags << 32; the first key of a branch page is empty (minus infinity)
      leaf node: data size = lo | hi << 16; flag 0x01 (F_BIGDATA): the data is the u64 number of an overflow page run whose
                 payload starts 16 bytes in; 0x02 / 0x04 (sub-databases, duplicates) are not used by the reference's stores
Keys compare as byte strings (memcmp, shorter first on a common prefix): LMDB's default comparator.
"""
import mmap
import os
import struct

MAGIC = 0xBEEFC0DE
P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
PAGEHDR = 16
INVALID = (1 << 64) - 1


class LmdbError(IOError):
    pass


class Environment:
    """`Environment(path)`: `path` is the environment directory (holding data.mdb) or the data file itself."""

    def __init__(self, path):
        self.path = os.path.join(path, 'data.mdb') if os.path.isdir(path) else path
        self._f = open(self.path, 'rb')
        size = os.fstat(self._f.fileno()).st_size
        if size < 2 * 512:
            raise LmdbError('%s: too small for an LMDB environment' % self.path)
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        m0 = self._meta(0)
        self.psize = m0['psize']
        if self.psize < 512 or self.psize & (self.psize - 1) or size < 2 * self.psize:
            raise LmdbError('%s: implausible page size %d' % (self.path, self.psize))
        m1 = self._meta(self.psize)
        self.meta = m1 if m1['txnid'] > m0['txnid'] else m0
        main = self.meta['main']
        if main['flags'] & 0x04:                       # MDB_DUPSORT
            raise LmdbError('duplicate-sorted main database: not a store the reference writes')
        if main['flags'] & (0x02 | 0x08):              # MDB_REVERSEKEY | MDB_INTEGERKEY: get() descends in memcmp key order only
            raise LmdbError('main database with MDB_REVERSEKEY / MDB_INTEGERKEY (flags 0x%x): keys are not in memcmp order' % main['flags'])
        self.entries, self.root, self.depth = main['entries'], main['root'], main['depth']

    def _meta(self, off):
        pgno, pad, flags, lower, upper = struct.unpack_from('<QHHHH', self._m, off)
        magic, version, address, mapsize = struct.unpack_from('<IIQQ', self._m, off + PAGEHDR)
        if not flags & P_META or magic != MAGIC:
            raise LmdbError('%s: no LMDB meta page at offset %d (magic %08x)' % (self.path, off, magic))
        if version != 1:
            raise LmdbError('%s: LMDB data format version %d (only 1 is known)' % (self.path, version))
        dbs = []
        for i in range(2):
            pad_, fl, depth, br, lf, ov, ent, root = struct.unpack_from('<IHHQQQQQ', self._m, off + PAGEHDR + 24 + 48 * i)
            dbs.append(dict(pad=pad_, flags=fl, depth=depth, entries=ent, root=root))
        last_pg, txnid = struct.unpack_from('<QQ', self._m, off + PAGEHDR + 24 + 96)
        return dict(psize=dbs[0]['pad'], main=dbs[1], last_pg=last_pg, txnid=txnid)

    # ---- pages and nodes
    def _page(self, pgno):
        off = pgno * self.psize
        if pgno == INVALID or off + PAGEHDR > len(self._m):
            raise LmdbError('page %d outside the file' % pgno)
        _, _, flags, lower, upper = struct.unpack_from('<QHHHH', self._m, off)
        return off, flags, (lower - PAGEHDR) >> 1

    def _node(self, page_off, i):
        noff = page_off + struct.unpack_from('<H', self._m, page_off + PAGEHDR + 2 * i)[0]
        lo, hi, flags, ksize = struct.unpack_from('<HHHH', self._m, noff)
        return noff, lo, hi, flags, ksize

    def _key(self, noff, ksize):
        return self._m[noff + 8:noff + 8 + ksize]

    def _value(self, noff, lo, hi, flags, ksize):
        if flags & (F_SUBDATA | F_DUPDATA):
            raise LmdbError('sub-database / duplicate node: not a store the reference writes')
        size = lo | (hi << 16)
        doff = noff + 8 + ksize
        if flags & F_BIGDATA:
            ov = struct.unpack_from('<Q', self._m, doff)[0]
            off, pflags, _ = self._page(ov)
            if not pflags & P_OVERFLOW:
                raise LmdbError('page %d is not an overflow page' % ov)
            doff = off + PAGEHDR
        if doff + size > len(self._m):
            raise LmdbError('value runs past the end of the file')
        return bytes(self._m[doff:doff + size])

    # ---- the operations the reference uses
    def stat(self):
        """`txn.stat()` of the main database: what lsun.py:27 reads is 'entries'."""
        return {'psize': self.psize, 'depth': self.depth, 'entries': self.entries}

    def get(self, key, default=None):
        """`txn.get(key)`: descend from the root by binary search on every page."""
        if self.root == INVALID:
            return default
        key = bytes(key)
        pgno = self.root
        for _ in range(64):
            off, flags, n = self._page(pgno)
            if flags & P_BRANCH:
                lo_i, hi_i = 1, n - 1                     # node 0 carries no key: it is the left-most child
                child = 0
                while lo_i <= hi_i:
                    mid = (lo_i + hi_i) >> 1
                    noff, _, _, _, ks = self._node(off, mid)
                    if self._key(noff, ks) <= key:
                        child, lo_i = mid, mid + 1
                    else:
                        hi_i = mid - 1
                noff, lo, hi, nflags, _ = self._node(off, child)
                pgno = lo | (hi << 16) | (nflags << 32)
                continue
            if not flags & P_LEAF:
                raise LmdbError('page %d is neither a branch nor a leaf' % pgno)
            lo_i, hi_i = 0, n - 1
            while lo_i <= hi_i:
                mid = (lo_i + hi_i) >> 1
                noff, lo, hi, nflags, ks = self._node(off, mid)
                k = self._key(noff, ks)
                if k == key:
                    return self._value(noff, lo, hi, nflags, ks)
                if k < key:
                    lo_i = mid + 1
                else:
                    hi_i = mid - 1
            return default
        raise LmdbError('tree deeper than 64 levels: corrupt environment')

    def items(self, values=True):
        """`txn.cursor()` order: every (key, value) of the main database in ascending key order (depth-first over the tree)."""
        if self.root == INVALID:
            return
        stack = [self.root]
        while stack:
            pgno = stack.pop()
            off, flags, n = self._page(pgno)
            if flags & P_BRANCH:
                kids = []
                for i in range(n):
                    _, lo, hi, nflags, _ = self._node(off, i)
                    kids.append(lo | (hi << 16) | (nflags << 32))
                stack.extend(reversed(kids))
            elif flags & P_LEAF:
                for i in range(n):
                    noff, lo, hi, nflags, ks = self._node(off, i)
                    k = bytes(self._key(noff, ks))
                    yield (k, self._value(noff, lo, hi, nflags, ks)) if values else k
            else:
                raise LmdbError('page %d is neither a branch nor a leaf' % pgno)

    def keys(self):
        return list(self.items(values=False))

    def close(self):
        try:
            self._m.close()
        finally:
            self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

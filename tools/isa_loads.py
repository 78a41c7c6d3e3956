#!/usr/bin/env python3
"""Memory-level-parallelism scan of a hipcc -S listing: per kernel, the order of vector-memory loads (L = dwordx4, l = narrower),
stores (S) and s_waitcnt vmcnt(n) (Wn) -- a kernel whose loads read 'L W0 L W0 ...' has ONE load in flight per wavefront.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude -o /tmp/x.s diff-pruning_amd/csrc/norm.hip
    python tools/isa_loads.py /tmp/x.s [kernel-name-substring]"""
import re
import sys


def scan(path, want=None):
    lines = open(path).read().split('\n')
    out = {}
    name = None
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        if name is None:
            continue
        t = l.strip()
        if t.startswith('s_endpgm'):
            name = None
        elif re.match(r'(global|buffer|flat)_load', t):
            out[name].append('B' if ' lds' in t else ('L' if 'dwordx4' in t else 'l'))
        elif re.match(r'(global|buffer|flat)_store', t):
            out[name].append('S')
        elif t.startswith('s_waitcnt') and 'vmcnt' in t:
            out[name].append('W' + re.search(r'vmcnt\((\d+)\)', t).group(1))
        elif t.startswith('s_cbranch'):
            out[name].append('.')
    for k, seq in out.items():
        if want and want not in k:
            continue
        s = ''.join(x if len(x) == 1 else '[%s]' % x for x in seq)
        s = re.sub(r'\.+', '.', s)
        loads = sum(1 for x in seq if x in 'Ll')
        w0 = sum(1 for x in seq if x == 'W0')
        print('%-60s loads %3d  vmcnt(0) %3d\n    %s' % (k[:60], loads, w0, s[:400]))


if __name__ == '__main__':
    scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files into {kernel: {counter: {avg_kb|avg, n}}} (per-launch averages).
    python tools/pmc_aggregate.py [--config NAME] OUT.json DIR [DIR ...]
--config: the bench.py --config the passes ran (default cifar256); stamped as `_config`, bench.py only quotes `roofline.traffic`
from a file whose `_config` is the config it is timing.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB (MI355X_MICROARCH.md, HBM section) -> key 'avg_kb'."""
import collections, csv, glob, json, os, re, sys


def clean(name):
    name = re.sub(r'^void\s+', '', name)
    return re.sub(r'\((dp_\w+|[\w\s\*,:<>]+)?\)\s*$', '', name).strip()


def main():
    argv = sys.argv[1:]
    config = 'cifar256'
    if argv and argv[0] == '--config':
        config, argv = argv[1], argv[2:]
    out, dirs = argv[0], argv[1:]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            per_dispatch = collections.defaultdict(float)
            names = {}
            for r in csv.DictReader(open(f)):
                key = (r['Dispatch_Id'], r['Counter_Name'])
                per_dispatch[key] += float(r['Counter_Value'])       # summed over XCDs / instances
                names[r['Dispatch_Id']] = clean(r['Kernel_Name'])
            for (disp, ctr), v in per_dispatch.items():
                a = agg[names[disp]][ctr]
                a[0] += 1
                a[1] += v
    res = {}
    for k, cs in agg.items():
        if k.startswith('at::') or len(k) > 160:
            continue
        res[k] = {c: {('avg_kb' if c.endswith('_SIZE') else 'avg'): v / n, 'n': n} for c, (n, v) in cs.items()}
    # stamp the source the counters were measured on: bench.py refuses the numbers once csrc/gemm.hip, csrc/winograd.hip or csrc/winograd2d.hip has changed
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = b''.join(open(os.path.join(root, 'diff-pruning_amd', 'csrc', f), 'rb').read() for f in ('gemm.hip', 'winograd.hip', 'winograd2d.hip', 'winograd2d_kloop.inc', 'wgrad2d.hip', 'wgrad2d_kloop.inc'))
    res['_gemm_hip_blob'] = hashlib.sha1(b'blob %d\0' % len(src) + src).hexdigest()
    res['_config'] = config
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, len(res), 'kernels')


if __name__ == '__main__':
    main()

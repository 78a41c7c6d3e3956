"""Golden vector for the FFHQ 90 / 10 split: executes the reference's own source lines (ddpm_exp/datasets/__init__.py:166-177, read
from the reference file at generation time) for a few dataset sizes and records the index lists (small N) or their digests (70 000,
the real FFHQ).  Run in the build container: python tests/golden/make_golden_ffhq_split.py"""
import hashlib
import json
import os
import textwrap

import numpy as np

REF = '/root/reference/ddpm_exp/datasets/__init__.py'
lines = open(REF).read().split('\n')
start = next(i for i, l in enumerate(lines) if 'num_items = len(dataset)' in l and i > 150)
end = next(i for i in range(start, len(lines)) if 'test_dataset = Subset(dataset, test_indices)' in lines[i])
src = textwrap.dedent('\n'.join(lines[start:end]))
out = {'source_lines': [start + 1, end], 'cases': []}
for n in (3, 10, 1000, 70000):
    np.random.seed(12345)                       # a caller state that must survive the split
    before = np.random.get_state()[1][:4].tolist()
    env = {'dataset': range(n), 'np': np}
    exec(src, env)
    after = np.random.get_state()[1][:4].tolist()
    assert before == after
    tr, te = env['train_indices'], env['test_indices']
    case = {'n': n, 'n_train': len(tr), 'n_test': len(te),
            'train_sha256': hashlib.sha256(np.asarray(tr, dtype=np.int64).tobytes()).hexdigest(),
            'test_sha256': hashlib.sha256(np.asarray(te, dtype=np.int64).tobytes()).hexdigest()}
    if n <= 10:
        case['train'], case['test'] = [int(i) for i in tr], [int(i) for i in te]
    out['cases'].append(case)
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ffhq_split.json'), 'w') as f:
    json.dump(out, f, indent=1)
print(json.dumps(out['cases'][:2]))

"""ORACLE -- test infrastructure only.

CPU restatement (plain PyTorch fp32) of the reference's Taylor-importance hot path.  Importable ONLY
from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product package
(diff-pruning_amd/) must never import it.  Pinned against golden vectors generated from the
reference itself (tests/golden/make_golden.py).
"""

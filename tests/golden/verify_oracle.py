"""Build-container check (imports /root/reference/src, like make_golden.py; never run by the tests):
the oracle's Laplacian builder against the reference's on random graphs — same generator as
tests/fuzz_parity.py — beyond the committed golden cases.

    python tests/golden/verify_oracle.py --cases 300
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, HERE)
from make_golden import _import_reference            # noqa: E402
from oracle import magnet_oracle as O                # noqa: E402


def graph(rng):
    n = int(rng.choice([1, 2, 3, 7, 50, 300, 2000, 20000]))
    e = int(rng.integers(0, 30 * n + 1)) if n > 1 else int(rng.integers(0, 3))
    row = rng.integers(0, n, e); col = rng.integers(0, n, e)
    if e > 4:
        k = e // 4
        row[:k], col[:k] = col[k:2 * k].copy(), row[k:2 * k].copy()
        row[2 * k:2 * k + k // 2] = row[:k // 2]; col[2 * k:2 * k + k // 2] = col[:k // 2]
    kind = int(rng.integers(0, 2))
    w = None if kind == 0 else rng.integers(1, 9, e).astype(np.float32)
    q = float(rng.choice([0.0, 0.05, 0.1, 0.25, 0.5, rng.random() * 0.5]))
    lam = float(rng.choice([2.0, 2.0, 3.0, 1.5, 2.1]))
    gcn = bool(rng.integers(0, 4) == 0)
    return row, col, n, w, q, lam, gcn


if __name__ == '__main__':
    ap = argparse.ArgumentParser(); ap.add_argument('--cases', type=int, default=200); ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    ref_h = _import_reference()[0]
    bad = 0
    for i in range(a.cases):
        row, col, n, w, q, lam, gcn = graph(np.random.default_rng([a.seed, i]))
        R = ref_h(row, col, n, q, True, True, lam, gcn, w); R.sort_indices()
        L = O.hermitian_decomp_sparse(row, col, n, q, max_eigen=lam, gcn_appr=gcn, edge_weight=w); L.sort_indices()
        ok = np.array_equal(R.indptr, L.indptr) and np.array_equal(R.indices, L.indices) and R.dtype == L.dtype
        ok = ok and (R.nnz == 0 or float(np.abs(R.data - L.data).max()) <= 2e-7 * max(float(np.abs(R.data).max()), 1e-30))
        if not ok:
            bad += 1
            print('case %d differs: n=%d e=%d q=%r lam=%r gcn=%d weighted=%d nnz %d / %d' % (i, n, len(row), q, lam, gcn, w is not None, R.nnz, L.nnz))
    print('%d cases, %d differences' % (a.cases, bad))
    sys.exit(1 if bad else 0)

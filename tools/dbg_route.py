"""Leaf arrays of the two routing forms on a few configurations (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util_data import clustered
from tests.gpu_util import make_builder
from pynndescent_amd import _capi

for metric, n, d, T in [("euclidean", 300000, 128, 8), ("euclidean", 300000, 128, 1), ("euclidean", 300000, 128, 2), ("euclidean", 150000, 128, 8),
                        ("euclidean", 300000, 64, 8), ("euclidean", 600000, 128, 4), ("cosine", 300000, 128, 8)]:
    x = clustered(n, d, 16, 64, seed=23)
    L = []
    for flags in (0, _capi.NND_FLAG_TEST_ROUTE_PLAIN):
        b = make_builder(x, metric, k=15, n_trees=T, flags=flags)
        b.make_forest()
        L.append(b.leaf_array())
        b.close()
    same = L[0].shape == L[1].shape
    diff = int((L[0] != L[1]).sum()) if same else -1
    rows = int((L[0] != L[1]).any(axis=1).sum()) if same else -1
    print(metric, n, d, T, "shapes", L[0].shape, L[1].shape, "diff elements", diff, "leaves", rows, flush=True)
    if same and rows:
        r = np.nonzero((L[0] != L[1]).any(axis=1))[0][:3]
        for q in r:
            a, c = set(L[0][q][L[0][q] >= 0].tolist()), set(L[1][q][L[1][q] >= 0].tolist())
            print("   leaf", q, "only coherent", sorted(a - c)[:6], "only plain", sorted(c - a)[:6], "sizes", len(a), len(c))

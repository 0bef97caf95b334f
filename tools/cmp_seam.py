"""Which half of the build is behind a recall gap to the CPU oracle?  Same points, four builds:
  A  GPU forest -> GPU descent      (the product)          B  ORACLE forest -> GPU descent  (leaf_array seam)
  C  GPU forest -> ORACLE descent   (leaf_array seam)      D  oracle forest -> oracle descent (the reference algorithm)
usage: python tools/cmp_seam.py [c3|c2] [oracle threads]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle as O
from tests.test_gpu_fullsize import _gen
from tests.gpu_util import make_builder

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n, d, latent, seed, metric, k, T, nonneg = {"c3": (1_200_000, 100, 24, 2, "cosine", 15, 12, False), "c2": (1_000_000, 128, 16, 1, "euclidean", 15, 8, True)}[cfg]
xh = _gen(n, d, latent, seed, torch.device("cuda", 0), nonneg).cpu().numpy()
rows = np.random.RandomState(5).choice(n, 2000, replace=False)
ti, _ = O.brute_force_knn(xh, 10, metric, rows=rows, kind="fast")
rec = lambda idx: round(O.recall(ti, idx[rows]), 4)
out = {"config": cfg}


def gpu_descent(b):
    b.init_random()
    cs = []
    for it in range(O.default_n_iters(n)):
        c = b.descent_iter()
        cs.append(int(c))
        if c <= 0.001 * k * n:
            break
    return b.finalize()[0], cs


b = make_builder(xh, metric, k=k, n_trees=T)
b.make_forest()
la_g = b.leaf_array()
b.init_from_leaves()
out["seeded_gpu_forest"] = rec(b.graph()[0])
idx, cs = gpu_descent(b)
out["A_gpu_forest_gpu_descent"] = {"recall": rec(idx), "updates": cs}
rng_state, _, ts = O.draw_rng_states(1, T)
la_o = O.make_leaf_array(xh, T, O.default_leaf_size(k), ts, metric == "cosine")
b.reset_graph()
b.init_from_leaf_array(la_o)
out["seeded_oracle_forest"] = rec(b.graph()[0])
idx, cs = gpu_descent(b)
out["B_oracle_forest_gpu_descent"] = {"recall": rec(idx), "updates": cs}
b.close()
lib = O.load("fast")
for name, la in (("C_gpu_forest_oracle_descent", la_g), ("D_oracle_forest_oracle_descent", la_o)):
    oi, _, tr = O.nn_descent(xh, k, rng_state, min(60, k), metric, O.default_n_iters(n), 0.001, la, n_threads=thr, lib=lib, return_trace=True)
    out[name] = {"recall": rec(oi), "updates": [int(v) for v in tr["c"]]}
print(json.dumps(out))

"""GPU forest vs the reference algorithm's forest (oracle) on the same points: leaf-size distribution and recall of the
k-lists right after leaf seeding (same seeding kernel).  usage: python tools/cmp_forest.py [n] [d] [metric]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests.util_data import clustered
from tests.test_gpu_kernels import make_builder

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 48
metric = sys.argv[3] if len(sys.argv) > 3 else "euclidean"
k, T = 15, 8
x = clustered(n, d, 10, 400, seed=17)
rows = np.random.RandomState(0).choice(n, 2000, replace=False)
ti, _ = O.brute_force_knn(x, 11, metric, rows=rows, kind="fast")
b = make_builder(x, metric, k=k, n_trees=T)
b.make_forest()
la_g = b.leaf_array()
b.init_from_leaves()
g_idx, _, _ = b.graph()
_, _, ts = O.draw_rng_states(1, T)
la_o = O.make_leaf_array(x, T, O.default_leaf_size(k), ts, metric == "cosine")
b.reset_graph()
b.init_from_leaf_array(la_o)
o_idx, _, _ = b.graph()


def desc(la):
    s = (la >= 0).sum(1)
    s = s[s > 0]
    return {"leaves": int(s.shape[0]), "mean": round(float(s.mean()), 2), "pairs_per_point": round(float((s * (s - 1.0)).sum() / s.sum()), 2),
            "pct": [int(v) for v in np.percentile(s, [1, 10, 25, 50, 75, 90, 99])]}


print(json.dumps({"n": n, "d": d, "metric": metric, "n_cells": b.stats()["n_cells"], "gpu": desc(la_g), "oracle": desc(la_o),
                  "recall_after_seeding": {"gpu_forest": round(O.recall(ti, g_idx[rows]), 4), "oracle_forest": round(O.recall(ti, o_idx[rows]), 4)}}))
b.close()

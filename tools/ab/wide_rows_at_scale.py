"""Rows of 200 / 256 neighbours at a size where the by-cell forest, the bucketed sampling with its large sub-regions and the join
sub-steps all run at scale: 150 000 x 32 clustered points, one GPU and two ranks on one GPU; recall@k on 1 000 rows against brute
force, iterations, wall time.  usage: python tools/ab/wide_rows_at_scale.py [n] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from pynndescent_amd import NNDescent, sharded
from tests.util_data import clustered

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x = clustered(n, 32, 12, 200, seed=4)
rows = np.random.RandomState(0).choice(n, 1000, replace=False)
ti, _ = O.brute_force_knn(x, k, "euclidean", rows=rows, kind="fast")
for rep in range(2):
    t = time.time()
    index = NNDescent(x, "euclidean", n_neighbors=k, n_trees=4, random_state=1)
    idx, dist = index._neighbor_graph
    dt = time.time() - t
print("one GPU   n=%d k=%d: recall@k %.4f, %d iterations, %.1f ms (class API, second call)" % (n, k, O.recall(ti, idx[rows]), index._build_stats["n_iters_run"], dt * 1e3))
assert (idx >= 0).all() and np.all(np.diff(dist, axis=1) >= 0)
t = time.time()
idx2, dist2, st, info = sharded.build_multi(x, 2, devices=[0, 0], metric="euclidean", n_neighbors=k, n_trees=4, seed=1)
print("two ranks n=%d k=%d: recall@k %.4f, %d iterations, %.1f ms, forest by cell: %s" % (n, k, O.recall(ti, idx2[rows]), info["iters"], (time.time() - t) * 1e3, info.get("forest_by_cell")))
assert (idx2 >= 0).all() and np.all(np.diff(dist2, axis=1) >= 0)

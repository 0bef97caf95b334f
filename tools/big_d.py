"""Large dimensions through the class API (d = 1000 .. 8192): recall@10 against exact neighbours.  usage: big_d.py"""
import os, sys, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import oracle as O
from pynndescent_amd import NNDescent
rs = np.random.RandomState(1)
for n, d, k, metric in [(3000, 1000, 15, "euclidean"), (3000, 2048, 15, "cosine"), (2000, 4099, 10, "euclidean"), (1500, 8192, 30, "cosine"), (20000, 1536, 15, "cosine")]:
    z = rs.standard_normal((n, 12)).astype(np.float32) @ rs.standard_normal((12, d)).astype(np.float32) + 0.1 * rs.standard_normal((n, d)).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        idx, dist = NNDescent(z, metric, n_neighbors=k, random_state=1).neighbor_graph
    ti, td = O.brute_force_knn(z, 10, metric, kind="fast")
    print(n, d, k, metric, "recall@10 %.4f" % O.recall(ti, idx), "dist finite", bool(np.isfinite(dist).all()), flush=True)

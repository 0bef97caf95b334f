"""Where the wall time of NNDescent(x, ...).neighbor_graph goes at 1 M x 128 (host side of the drop-in call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import sift_like
import torch
from pynndescent_amd import NNDescent, _capi

x = sift_like(1_000_000, 128, seed=1, device=torch.device("cuda", 0), sample_seed=100).cpu().numpy()
def T(f, rep=3):
    best = 1e9
    for _ in range(rep):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
for _ in range(2):
    t_ctor, idx = T(lambda: NNDescent(x, "euclidean", n_neighbors=15, n_trees=8, random_state=1), rep=2)
    print("constructor %.1f ms" % t_ctor)
    t_ng, g = T(lambda: idx.neighbor_graph)
    print("neighbor_graph property %.1f ms" % t_ng)
    i0, d0 = idx._neighbor_graph
    print("  ndarray.copy %.1f ms | host_copy %.1f ms" % (T(lambda: i0.copy())[0], T(lambda: _capi.host_copy(i0))[0]))
    print("  np.sqrt %.1f ms | host_sqrt %.1f ms" % (T(lambda: np.sqrt(d0))[0], T(lambda: _capi.host_sqrt(d0))[0]))
    print("  np.empty + fill %.1f ms" % T(lambda: np.empty_like(i0).fill(0))[0])
    t_all, _ = T(lambda: NNDescent(x, "euclidean", n_neighbors=15, n_trees=8, random_state=1).neighbor_graph, rep=2)
    print("NNDescent(...).neighbor_graph %.1f ms" % t_all)

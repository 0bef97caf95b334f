"""Forest stage with the coherent routing passes vs the plain walk (NND_FLAG_TEST_ROUTE_PLAIN), same points, same seeds.
usage: python tools/ab/ab_forest.py [n] [n_trees] [reps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pynndescent_amd import _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
xd = bench.sift_like(n, 128, seed=1, device="cuda:0", sample_seed=100)
torch.cuda.synchronize()
for name, flags in (("coherent", 0), ("plain", _capi.NND_FLAG_TEST_ROUTE_PLAIN), ("coherent", 0)):
    b = _capi.Builder(n=n, dim=128, metric=0, n_neighbors=15, n_trees=T, leaf_size=75, max_depth=200, max_candidates=15, n_iters=5,
                      delta=0.001, rng_state=(1, 2, 3), tree_rng=(4, 5, 6), device=0, flags=flags)
    b.set_data_device(xd.data_ptr())
    ms = []
    for r in range(reps):
        b.make_forest()
        ms.append(round(b.stats()["ms_forest"], 3))
    print(json.dumps({"route": name, "n": n, "trees": T, "ms_forest": ms, "n_cells": b.stats()["n_cells"], "leaves": b.stats()["n_leaves"]}))
    b.close()
    _capi.load_library().nnd_release_pending()

"""Per-iteration stage times of one build (stats arrays): python tools/iter_stats.py [n] [n_trees] [k]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from pynndescent_amd import _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
x = bench.sift_like(n, 128, seed=1, device="cuda:0", sample_seed=100)
torch.cuda.synchronize()
b = _capi.Builder(n=n, dim=128, metric=0, n_neighbors=k, n_trees=T, leaf_size=max(60, min(256, 5 * k)), max_depth=200, max_candidates=min(60, k),
                  n_iters=max(5, int(round(np.log2(n)))), delta=0.001, rng_state=(1, 2, 3), tree_rng=(4, 5, 6), device=0)
b.set_data_device(x.data_ptr(), keepalive=x)
oi = torch.empty((n, k), dtype=torch.int32, device="cuda:0")
od = torch.empty((n, k), dtype=torch.float32, device="cuda:0")
for _ in range(3):
    b.build_device(oi.data_ptr(), od.data_ptr())
    b.synchronize()
st = b.stats()
it = st["n_iters_run"]
out = {"n": n, "trees": T, "k": k, "iters": it}
for name in ("ms_sample", "ms_join", "ms_merge", "updates", "join_pairs", "join_active", "proposals"):
    out[name] = [round(float(v), 3) for v in st[name][:it]]
for name in ("ms_prep", "ms_forest", "ms_leaf_init", "ms_random_init", "ms_descent", "ms_finalize"):
    out[name] = round(float(st[name]), 3)
print(json.dumps(out))

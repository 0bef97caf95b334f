"""What would renumbering the points into the first tree's leaf order buy?  The same set built as given and after a
permutation into the leaf order of one RP tree (done here, outside the library: the best case for every gather).
usage: python tools/ab/ab_permute.py [n] [n_trees]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from pynndescent_amd import _capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
x = bench.sift_like(n, 128, seed=1, device="cuda:0", sample_seed=100)
torch.cuda.synchronize()


def builder(trees):
    return _capi.Builder(n=n, dim=128, metric=0, n_neighbors=15, n_trees=trees, leaf_size=75, max_depth=200, max_candidates=15,
                         n_iters=max(5, int(round(np.log2(n)))), delta=0.001, rng_state=(1, 2, 3), tree_rng=(4, 5, 6), device=0)


def run(xd, name):
    b = builder(T)
    b.set_data_device(xd.data_ptr(), keepalive=xd)
    oi = torch.empty((n, 15), dtype=torch.int32, device="cuda:0")
    od = torch.empty((n, 15), dtype=torch.float32, device="cuda:0")
    for _ in range(3):
        b.build_device(oi.data_ptr(), od.data_ptr())
        b.synchronize()
    st = b.stats()
    ms = {k: round(float(np.sum(v)), 2) for k, v in st.items() if k.startswith("ms_")}
    print(json.dumps({"order": name, "n": n, "trees": T, "iters": st["n_iters_run"], **ms}), flush=True)
    b.close()
    _capi.load_library().nnd_release_pending()


run(x, "as given")
b1 = builder(1)
b1.set_data_device(x.data_ptr())
b1.make_forest()
la = b1.leaf_array()
b1.close()
order = la.reshape(-1)
order = order[order >= 0]
assert order.shape[0] == n and np.unique(order).shape[0] == n
x2 = x[torch.from_numpy(order.astype(np.int64)).cuda()].contiguous()
torch.cuda.synchronize()
run(x2, "one tree's leaf order")
run(x, "as given")

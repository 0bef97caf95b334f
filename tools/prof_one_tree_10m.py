"""10 M points, ONE tree, forest + leaf seeding only -- what a rank of the 8-way sharded build does per local tree
(run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import sift_like
from pynndescent_amd import _capi
n, d, k, T = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 128, 15, int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
x = sift_like(n, d, seed=1, device=dev, sample_seed=100)
rs = np.random.RandomState(1234); lim = np.iinfo(np.int32)
rng = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64); ts = rs.randint(lim.min + 1, lim.max - 1, size=(max(T, 1), 3)).astype(np.int64)
b = _capi.Builder(n, d, 0, k, T, 75, 200, k, 0, 0.001, rng, ts[0])   # n_iters = 0: forest, leaf seeding, random fill, finalize
oi = torch.empty((n, k), dtype=torch.int32, device=dev); od = torch.empty((n, k), dtype=torch.float32, device=dev)
for _ in range(3):
    b.set_data_device(x.data_ptr(), keepalive=x); b.build_device(oi.data_ptr(), od.data_ptr())
b.synchronize()
st = b.stats()
print({kk: round(st[kk], 2) for kk in ("ms_prep", "ms_forest", "ms_leaf_init", "ms_random_init", "ms_finalize")})

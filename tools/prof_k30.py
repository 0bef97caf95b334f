"""k = 30 build at 1 M points, three times (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import sift_like
from pynndescent_amd import _capi
n, d, k, T = 1_000_000, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 30, 8
dev = torch.device("cuda", 0)
x = sift_like(n, d, seed=1, device=dev, sample_seed=100)
rs = np.random.RandomState(1234); lim = np.iinfo(np.int32)
rng = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64); ts = rs.randint(lim.min + 1, lim.max - 1, size=(T, 3)).astype(np.int64)
b = _capi.Builder(n, d, 0, k, T, max(60, min(256, 5 * k)), 200, min(60, k), 20, 0.001, rng, ts[0])
oi = torch.empty((n, k), dtype=torch.int32, device=dev); od = torch.empty((n, k), dtype=torch.float32, device=dev)
for _ in range(3):
    b.set_data_device(x.data_ptr(), keepalive=x); b.build_device(oi.data_ptr(), od.data_ptr())
b.synchronize()
st = b.stats()
print({kk: (round(st[kk], 2) if isinstance(st[kk], float) else st[kk]) for kk in ("n_iters_run", "ms_forest", "ms_leaf_init", "ms_finalize", "proposals", "updates", "join_pairs", "join_rows")},
      "join", [round(v, 2) for v in st["ms_join"]], "sample", [round(v, 2) for v in st["ms_sample"]], "merge", [round(v, 2) for v in st["ms_merge"]])

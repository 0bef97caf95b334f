"""GPU build vs CPU oracle on the same points: iterations run, update counts per iteration, recall@10 of both on the same
sample (where does a recall gap come from: fewer iterations, fewer updates, or the same counts and worse lists?).
usage: python tools/cmp_trace.py [c3|c2|3m] [oracle threads]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle as O
from tests.test_gpu_fullsize import _gen, _build

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n, d, latent, seed, metric, k, T, nonneg = {"c3": (1_200_000, 100, 24, 2, "cosine", 15, 12, False), "c2": (1_000_000, 128, 16, 1, "euclidean", 15, 8, True),
                                             "3m": (3_000_000, 128, 16, 3, "euclidean", 15, 10, True)}[cfg]
x = _gen(n, d, latent, seed, torch.device("cuda", 0), nonneg)
idx, dist, st = _build(x, metric, k, T)
xh = x.cpu().numpy()
t0 = time.time()
oidx, _, tr = O.build_index(xh, metric, n_neighbors=k, n_trees=T, random_state=1, n_threads=thr, kind="fast", return_trace=True)
t_or = time.time() - t0
rows = np.random.RandomState(5).choice(n, 2000, replace=False)
ti, _ = O.brute_force_knn(xh, 10, metric, rows=rows, kind="fast")
it = st["n_iters_run"]
print(json.dumps({"config": cfg, "gpu": {"iters": it, "updates": [int(v) for v in st["updates"][:it]], "recall": round(O.recall(ti, idx.cpu().numpy()[rows]), 4)},
                  "oracle": {"iters": tr["iters"], "c": [int(v) for v in tr["c"]], "recall": round(O.recall(ti, oidx[rows]), 4), "seconds": round(t_or, 1)},
                  "stop_threshold": 0.001 * k * n}))

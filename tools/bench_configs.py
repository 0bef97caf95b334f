"""Build time / recall of the other BASELINE configurations on one MI355X (info lines for profiles/, not the bench line).
usage: python tools/bench_configs.py [c3 c4 c5 ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from oracle import oracle as O
from pynndescent_amd import _capi

CONFIGS = {
    # name: (n, d, latent, seed, metric, k, n_trees, nonneg)
    "c2": (1_000_000, 128, 16, 1, "euclidean", 15, 8, True),
    "c3": (1_200_000, 100, 24, 2, "cosine", 15, 12, False),
    "c4_one_gpu": (10_000_000, 128, 16, 3, "euclidean", 15, 12, True),
    "c5": (290_000, 256, 32, 4, "cosine", 15, 11, False),
    "k30": (1_000_000, 128, 16, 1, "euclidean", 30, 8, True),
    "k60": (500_000, 128, 16, 1, "euclidean", 60, 8, True),
    "s1m": (1_000_000, 128, 16, 3, "euclidean", 15, 12, True),
    "s2m": (2_000_000, 128, 16, 3, "euclidean", 15, 12, True),
    "s4m": (4_000_000, 128, 16, 3, "euclidean", 15, 12, True),
}


def gen(n, d, latent, seed, dev, nonneg):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    centres = torch.randn(1024, latent, generator=g, device=dev) * 3.0
    proj = torch.randn(latent, d, generator=g, device=dev) / latent ** 0.5
    assign = torch.randint(0, 1024, (n,), generator=g, device=dev)
    x = (centres[assign] + torch.randn(n, latent, generator=g, device=dev)) @ proj
    x = x + 0.3 * torch.randn(n, d, generator=g, device=dev)
    if nonneg:
        x = (x + 12.0).clamp_min(0) * 9.0
    return x.contiguous()


def exact_top10(x, rows, metric):
    """Exact 10-NN of the sampled rows: f32 Gram pre-selection in chunks of 1M columns (torch.topk over rows of several
    million columns returned wrong answers on this stack), 64 best candidates refined in float64."""
    n = x.shape[0]
    if metric == "euclidean":
        xc = x - x.mean(0, keepdim=True)
    else:
        xc = x / x.norm(dim=1, keepdim=True)
    q = xc[rows]
    best_v = best_i = None
    for c0 in range(0, n, 1_000_000):
        xs = xc[c0:c0 + 1_000_000]
        if metric == "euclidean":
            dch = (q * q).sum(1, keepdim=True) + (xs * xs).sum(1)[None, :] - 2.0 * (q @ xs.T)
        else:
            dch = 1.0 - q @ xs.T
        tk = dch.topk(min(32, xs.shape[0]), dim=1, largest=False)
        best_v = tk.values if best_v is None else torch.cat([best_v, tk.values], 1)
        best_i = tk.indices + c0 if best_i is None else torch.cat([best_i, tk.indices + c0], 1)
    o = best_v.argsort(dim=1)[:, :64]
    cand = torch.gather(best_i, 1, o)
    qd = x[rows].double()
    nb = x[cand].double()
    if metric == "euclidean":
        dd = ((qd[:, None, :] - nb) ** 2).sum(-1)
    else:
        dd = 1.0 - (qd[:, None, :] * nb).sum(-1) / (qd.norm(dim=1)[:, None] * nb.norm(dim=2))
    return torch.gather(cand, 1, dd.argsort(dim=1)[:, :10])


def run(name):
    n, d, latent, seed, metric, k, n_trees, nonneg = CONFIGS[name]
    dev = torch.device("cuda", 0)
    x = gen(n, d, latent, seed, dev, nonneg)
    rng_state, _, ts = O.draw_rng_states(1, n_trees)
    b = _capi.Builder(n, d, O.METRICS[metric], k, n_trees, O.default_leaf_size(k), 200, min(60, k), O.default_n_iters(n), 0.001,
                      rng_state, ts[0])
    idx = torch.empty((n, k), dtype=torch.int32, device=dev)
    dist = torch.empty((n, k), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()  # x is produced on torch's stream; the library reads it on its own stream
    b.set_data_device(x.data_ptr(), keepalive=x)
    times = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.build_device(idx.data_ptr(), dist.data_ptr())
        b.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    st = b.stats()
    rows = torch.from_numpy(np.random.RandomState(0).choice(n, 500, replace=False)).to(dev)
    true10 = exact_top10(x, rows, metric)
    rec = bench.recall_at(true10, idx[rows], 10)
    b.close()
    out = {"config": name, "n": n, "d": d, "metric": metric, "k": k, "n_trees": n_trees, "ms_best": round(min(times), 2),
           "ms_all": [round(t, 2) for t in times], "points_per_s": round(n / (min(times) * 1e-3)), "recall_at_10": round(rec, 4),
           "iters": st["n_iters_run"], "tree_levels": st["tree_levels"],
           "stage_ms": {key: round(st[key], 2) for key in ("ms_prep", "ms_forest", "ms_leaf_init", "ms_finalize")},
           "ms_join": round(sum(st["ms_join"]), 2), "ms_sample": round(sum(st["ms_sample"]), 2), "ms_merge": round(sum(st["ms_merge"]), 2)}
    print(json.dumps(out))
    sys.stdout.flush()


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["c3", "c5", "c4_one_gpu"]):
        run(name)

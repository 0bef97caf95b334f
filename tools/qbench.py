"""Torch-free quick bench for kernel A/B work on the GPU box (no `import torch`: saves the 1-2 min first import).

    python tools/qbench.py [--n 1000000] [--reps 3] [--env NND_CELL_LEAF=190 --env NND_CELL_LEAF=48 ...] [--whole]

The SIFT-like stand-in of bench.py regenerated with numpy, one handle per configuration (environment variables are read
by nnd_create), `reps` builds through the stage entry points, stage timings of the last build (HIP events on the
library's stream), recall@10 on a sample against exact neighbours (CPU oracle brute force = checker only)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pynndescent_amd import _capi  # noqa: E402


def sift_like_np(n, d, seed=1, latent=16, n_clusters=1024, noise=0.3):
    rs = np.random.RandomState(seed)
    centres = rs.standard_normal((n_clusters, latent)).astype(np.float32) * 3.0
    proj = (rs.standard_normal((latent, d)) / latent ** 0.5).astype(np.float32)
    assign = rs.randint(0, n_clusters, n)
    z = centres[assign] + rs.standard_normal((n, latent)).astype(np.float32)
    x = z @ proj + noise * rs.standard_normal((n, d)).astype(np.float32)
    x = np.clip(x + 12.0, 0.0, None) * (218.0 / 24.0)
    return np.ascontiguousarray(x, np.float32)


def run(x, metric, k, n_trees, reps, env, check_rows=None, true_idx=None, leaf_check=False):
    for kv in env:
        key, val = kv.split("=", 1)
        os.environ[key] = val
    n, d = x.shape
    n_iters = max(5, int(round(np.log2(n))))
    b = _capi.Builder(n, d, metric, k, n_trees, max(60, min(256, 5 * k)), 200, min(60, k), n_iters, 0.001, (11, 22, 33), (44, 55, 66))
    b.set_data_host(x)
    out = None
    t_wall = []
    for _ in range(reps):
        t0 = time.perf_counter()
        b.reset_graph()
        b.make_forest()
        b.init_from_leaves()
        b.init_random()
        b.descent()
        idx, dist = b.finalize()
        t_wall.append(time.perf_counter() - t0)
        out = b.stats()
    rec = None
    if check_rows is not None:
        rec = float(np.mean([np.isin(t, a).sum() for t, a in zip(true_idx, idx[check_rows])]) / true_idx.shape[1])
    extra = {}
    if leaf_check:
        la = b.leaf_array()
        ids = la[la >= 0]
        cnt = np.bincount(ids, minlength=n)
        extra = {"leaf_rows": int(la.shape[0]), "every_point_once_per_tree": bool(np.all(cnt == n_trees)),
                 "max_leaf": int((la >= 0).sum(1).max()), "mean_leaf": float((la >= 0).sum(1).mean())}
    b.close()
    for kv in env:
        os.environ.pop(kv.split("=", 1)[0], None)
    res = {"env": env, "forest": round(out["ms_forest"], 3), "leaf_init": round(out["ms_leaf_init"], 3),
           "join": round(sum(out["ms_join"]), 3), "sample": round(sum(out["ms_sample"]), 3),
           "merge": round(sum(out["ms_merge"]), 3), "finalize": round(out["ms_finalize"], 3), "iters": out["n_iters_run"],
           "levels": out["tree_levels"], "cells": out["n_cells"], "leaves": out["n_leaves"], "recall": rec,
           "sum_stage_ms": round(out["ms_forest"] + out["ms_leaf_init"] + sum(out["ms_join"]) + sum(out["ms_sample"]) +
                                 sum(out["ms_merge"]) + out["ms_finalize"] + out["ms_random_init"], 3),
           "wall_ms_min": round(min(t_wall) * 1e3, 2)}
    res.update(extra)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--trees", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--latent", type=int, default=16)
    ap.add_argument("--metric", type=int, default=0)
    ap.add_argument("--env", action="append", default=[], help="KEY=VAL[,KEY=VAL...] one configuration per --env")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--leaf-check", action="store_true")
    ap.add_argument("--reorder", action="store_true", help="experiment: permute the input rows into the first tree's leaf order first")
    args = ap.parse_args()
    x = sift_like_np(args.n, args.d, latent=args.latent)
    if args.reorder:
        n_iters = max(5, int(round(np.log2(args.n))))
        b = _capi.Builder(args.n, args.d, args.metric, args.k, args.trees, max(60, min(256, 5 * args.k)), 200, min(60, args.k), n_iters,
                          0.001, (11, 22, 33), (44, 55, 66))
        b.set_data_host(x)
        b.make_forest()
        la = b.leaf_array()
        b.close()
        ids = la[la >= 0]
        order = ids[: args.n]  # leaves are listed tree by tree: the first n ids are tree 0, a permutation of all points
        assert np.array_equal(np.sort(order), np.arange(args.n))
        x = np.ascontiguousarray(x[order])
    rows = ti = None
    if not args.no_recall:
        from oracle import oracle as O  # checker only

        rows = np.random.RandomState(0).choice(args.n, 1000, replace=False)
        ti, _ = O.brute_force_knn(x, 10, "euclidean" if args.metric == 0 else "cosine", rows=rows)
    configs = [e.split(",") if e else [] for e in (args.env or [""])]
    for env in configs:
        print(json.dumps(run(x, args.metric, args.k, args.trees, args.reps, env, rows, ti, args.leaf_check)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()

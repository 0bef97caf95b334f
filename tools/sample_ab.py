"""Candidate sampling, bucketed transposition (round 5) vs one global atomicMin per edge (rounds 1-4): per-iteration stage
time on C2' (1 M x 128) and recall@10 on C3' (1.2 M x 100 cosine, 20 000 rows, several seeds) -- the rows, generator and
seeds of tools/recall_study.py, whose oracle figures (0.98996 / 0.99024 for oracle seeds 1 / 2) therefore apply.
usage: python tools/sample_ab.py [timing] [recall] [n_rows] [gpu seeds]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from oracle import oracle as O
from pynndescent_amd import _capi
from tools.bench_configs import CONFIGS, gen, exact_top10

what = [a for a in sys.argv[1:] if not a.isdigit()] or ["timing", "recall"]
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
n_rows = nums[0] if nums else 20000
g_seeds = nums[1] if len(nums) > 1 else 4
dev = torch.device("cuda", 0)
VARIANTS = [("bucketed", 0), ("atomic", _capi.NND_FLAG_TEST_SAMPLE_ATOMIC)]


def build(x, cfg, seed, flags, idx, dist):
    n, d, latent, _, metric, k, T, nonneg = CONFIGS[cfg]
    rng_state, _, ts = O.draw_rng_states(seed, T)
    b = _capi.Builder(n, d, O.METRICS[metric], k, T, O.default_leaf_size(k), 200, min(60, k), O.default_n_iters(n), 0.001, rng_state, ts[0], flags=flags)
    torch.cuda.synchronize()  # x (and whatever else torch still has in flight) must be complete: the library runs on its own stream
    b.set_data_device(x.data_ptr(), keepalive=x)
    b.build_device(idx.data_ptr(), dist.data_ptr())
    b.synchronize()
    st = b.stats()
    b.close()
    return st


if "timing" in what:
    for cfg in ("c2",):
        n, d, latent, seed, metric, k, T, nonneg = CONFIGS[cfg]
        x = gen(n, d, latent, seed, dev, nonneg)
        idx = torch.empty((n, k), dtype=torch.int32, device=dev)
        dist = torch.empty((n, k), dtype=torch.float32, device=dev)
        rows = torch.from_numpy(np.random.RandomState(0).choice(n, 2000, replace=False)).to(dev)
        true10 = exact_top10(x, rows, metric)
        for name, flags in VARIANTS:
            best = None
            for rep in range(4):
                st = build(x, cfg, 1, flags, idx, dist)
                it = st["n_iters_run"]
                cur = {"ms_sample": [round(v, 3) for v in st["ms_sample"][:it]], "sample_total": round(sum(st["ms_sample"][:it]), 3),
                       "ms_descent": round(st["ms_descent"], 3), "iters": it, "updates": [int(v) for v in st["updates"][:it]]}
                if rep and (best is None or cur["sample_total"] < best["sample_total"]):
                    best = cur
            best["recall"] = round(bench.recall_at(true10, idx[rows], 10), 5)
            print(json.dumps({"config": cfg, "sampler": name, **best}), flush=True)
            _capi.load_library().nnd_release_pending()
        del x, idx, dist

if "recall" in what:
    cfg = "c3"
    n, d, latent, seed, metric, k, T, nonneg = CONFIGS[cfg]
    x = gen(n, d, latent, seed, dev, nonneg)
    rows = torch.from_numpy(np.random.RandomState(0).choice(n, n_rows, replace=False)).to(dev)
    true10 = torch.cat([exact_top10(x, rows[i:i + 2000], metric) for i in range(0, n_rows, 2000)])
    idx = torch.empty((n, k), dtype=torch.int32, device=dev)
    dist = torch.empty((n, k), dtype=torch.float32, device=dev)
    chk0 = (int(true10.sum()), int(rows.sum()), float(x.double().sum()))
    for name, flags in VARIANTS:
        rec, its = [], []
        for s in range(1, g_seeds + 1):
            st = build(x, cfg, s, flags, idx, dist)
            its.append(st["n_iters_run"])
            rec.append(bench.recall_at(true10, idx[rows], 10))
            chk = (int(true10.sum()), int(rows.sum()), float(x.double().sum()))
            if chk != chk0:
                print("CORRUPTED after", name, s, chk0, chk, flush=True)
                chk0 = chk
        print(json.dumps({"config": cfg, "sampler": name, "recall_mean": round(float(np.mean(rec)), 5), "recall_std": round(float(np.std(rec)), 5),
                          "recalls": [round(r, 5) for r in rec], "iters": its, "oracle_recalls_r04": [0.98996, 0.99024]}), flush=True)
        _capi.load_library().nnd_release_pending()

if "debug" in what:  # three consecutive builds of one seed per variant, with the iteration counters
    cfg = "c3"
    n, d, latent, seed, metric, k, T, nonneg = CONFIGS[cfg]
    x = gen(n, d, latent, seed, dev, nonneg)
    rows = torch.from_numpy(np.random.RandomState(0).choice(n, 4000, replace=False)).to(dev)
    true10 = torch.cat([exact_top10(x, rows[i:i + 2000], metric) for i in range(0, 4000, 2000)])
    idx = torch.empty((n, k), dtype=torch.int32, device=dev)
    dist = torch.empty((n, k), dtype=torch.float32, device=dev)
    for name, flags in VARIANTS:
        for rep in range(3):
            idx.fill_(-7)
            st = build(x, cfg, 1, flags, idx, dist)
            it = st["n_iters_run"]
            print(json.dumps({"lib": os.path.basename(_capi.LIB_PATH), "sampler": name, "rep": rep, "recall": round(bench.recall_at(true10, idx[rows], 10), 5),
                              "idx_min": int(idx.min()), "updates": [int(v) for v in st["updates"][:it]], "active": [int(v) for v in st["join_active"][:it]],
                              "proposals": [int(v) for v in st["proposals"][:it]], "ms_sample": [round(v, 3) for v in st["ms_sample"][:it]]}), flush=True)
        _capi.load_library().nnd_release_pending()

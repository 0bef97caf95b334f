"""Where does a recall gap in the MID regime come from?  200 000 iid Gaussian points (the reference algorithm lands at recall@10 of
0.6-0.9 there): GPU build against the CPU oracle on the same points -- several seeds each side, 4 000 sample rows; the recall after
1, 2, 3, ... iterations on both sides (n_iters forced, delta = 0); the GPU with sub-steps (join_blocks) as the reference's
16384-vertex blocks apply them (pynndescent_.py:239-261).
usage: python tools/mid_regime_study.py [d=32] [metric=euclidean] [n=200000]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle as O
from pynndescent_amd import _capi

d = int(sys.argv[1]) if len(sys.argv) > 1 else 32
metric = sys.argv[2] if len(sys.argv) > 2 else "euclidean"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000
k = 15
x = np.random.RandomState(3).standard_normal((n, d)).astype(np.float32)
xt = torch.from_numpy(x).cuda()
T = O.default_n_trees(n)
rows = np.random.RandomState(5).choice(n, 4000, replace=False)
ti, _ = O.brute_force_knn(x, 10, metric, rows=rows, kind="fast")
idx = torch.empty((n, k), dtype=torch.int32, device="cuda")
dist = torch.empty((n, k), dtype=torch.float32, device="cuda")


def gpu(seed, n_iters=None, delta=0.001, join_blocks=1, mc=None, flags=0):
    rng_state, _, ts = O.draw_rng_states(seed, T)
    b = _capi.Builder(n, d, O.METRICS[metric], k, T, O.default_leaf_size(k), 200, min(60, k) if mc is None else mc,
                      O.default_n_iters(n) if n_iters is None else n_iters, delta, rng_state, ts[0], join_blocks=join_blocks, flags=flags)
    b.set_data_device(xt.data_ptr(), keepalive=xt)
    if n_iters == 0:
        b.reset_graph(); b.make_forest(); b.init_from_leaves(); b.init_random(); b.finalize_device(idx.data_ptr(), dist.data_ptr())
    else:
        b.build_device(idx.data_ptr(), dist.data_ptr())
    b.synchronize()
    st = b.stats()
    b.close()
    it = st["n_iters_run"]
    return round(O.recall(ti, idx.cpu().numpy()[rows]), 4), it, [int(v) for v in st["updates"][:it]]


def orc(seed, n_iters=None, delta=0.001, thr=64):
    oi, _, tr = O.build_index(x, metric, n_neighbors=k, n_trees=T, random_state=seed, n_threads=thr, kind="fast", n_iters=n_iters, delta=delta, return_trace=True)
    return round(O.recall(ti, oi[rows]), 4), tr["iters"], [int(v) for v in tr["c"]]


print(json.dumps({"n": n, "d": d, "metric": metric, "trees": T, "n_iters_default": O.default_n_iters(n), "stop_threshold": 0.001 * k * n}), flush=True)
ors = []
for s in (1, 2, 3):
    o = orc(s)
    ors.append(o[0])
    print(json.dumps({"seed": s, "gpu_join_blocks_1": gpu(s), "oracle": o}), flush=True)
print(json.dumps({"oracle_recall_mean": round(float(np.mean(ors)), 4)}), flush=True)
if "--curves" in sys.argv:
    print(json.dumps({"oracle_threads_8": orc(1, thr=8), "oracle_threads_256": orc(1, thr=256)}), flush=True)
    for it in (0, 1, 2, 3, 5, 8, 12):
        print(json.dumps({"forced_iters": it, "gpu": gpu(1, n_iters=it, delta=0.0)[0], "oracle": orc(1, n_iters=it, delta=0.0)[0] if it > 0 else None}), flush=True)
for jb in (2, 4, 12):
    print(json.dumps({"join_blocks": jb, "gpu": gpu(1, join_blocks=jb)}), flush=True)
# join_blocks = 0: the library's schedule (sub-steps follow the update volume); under a KNOBS build of capi.hip
# (PYNND_AMD_LIB=pynndescent_amd/_exp/lib_kn.so) the schedule's parameters can be swept
for env in ({}, {"NND_JB_MAX": "1"}, {"NND_JB_DIV": "1"}, {"NND_JB_MAX": "16", "NND_JB_DIV": "1", "NND_JB_FIRST": "16"}, {"NND_JB_MAX": "16", "NND_JB_DIV": "1", "NND_JB_FIRST": "8"}):
    for kk in ("NND_JB_MAX", "NND_JB_DIV", "NND_JB_FIRST"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    rs = [gpu(s, join_blocks=0) for s in (1, 2, 3)]
    print(json.dumps({"auto_schedule": env, "recall_mean": round(float(np.mean([r[0] for r in rs])), 4), "runs": rs}), flush=True)

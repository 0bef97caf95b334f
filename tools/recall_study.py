"""Recall@10 of the GPU build under experiment knobs, several seeds each, on MANY sample rows (sigma of one number ~ 0.00025),
next to the CPU oracle on the same rows.  Needs a KNOBS build of capi.hip (PYNND_AMD_LIB=pynndescent_amd/_exp/lib_kn.so).
usage: python tools/recall_study.py [c3|c2] [n_rows] [oracle seeds] [gpu seeds]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from oracle import oracle as O
from pynndescent_amd import _capi
from tools.bench_configs import CONFIGS, gen, exact_top10

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
o_seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
g_seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
n, d, latent, seed, metric, k, T, nonneg = CONFIGS[cfg]
dev = torch.device("cuda", 0)
x = gen(n, d, latent, seed, dev, nonneg)
rows = torch.from_numpy(np.random.RandomState(0).choice(n, n_rows, replace=False)).to(dev)
true10 = torch.cat([exact_top10(x, rows[i:i + 2000], metric) for i in range(0, n_rows, 2000)])
idx = torch.empty((n, k), dtype=torch.int32, device=dev)
dist = torch.empty((n, k), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
KNOBS = ("NND_FOREST_WHOLE", "NND_SAMPLE_STRIDE", "NND_CELL_LEAF", "NND_RCAP", "NND_PCAP")
VARIANTS = [{}, {"NND_FOREST_WHOLE": "1"}, {"NND_SAMPLE_STRIDE": "64", "NND_CELL_LEAF": "6"}, {"NND_SAMPLE_STRIDE": "4", "NND_CELL_LEAF": "96"},
            {"NND_SAMPLE_STRIDE": "8", "NND_CELL_LEAF": "48"}, {"NND_RCAP": "64"}, {"NND_RCAP": "16"}, {"NND_PCAP": "32"}, {"NND_PCAP": "16"}]
if os.environ.get("STUDY_VARIANTS"):  # e.g. STUDY_VARIANTS=0,5,6
    VARIANTS = [VARIANTS[int(i)] for i in os.environ["STUDY_VARIANTS"].split(",")]
for env in VARIANTS:
    for kk in KNOBS:
        os.environ.pop(kk, None)
    os.environ.update(env)
    rec, its = [], []
    for s in range(1, g_seeds + 1):
        rng_state, _, ts = O.draw_rng_states(s, T)
        b = _capi.Builder(n, d, O.METRICS[metric], k, T, O.default_leaf_size(k), 200, min(60, k), O.default_n_iters(n), 0.001, rng_state, ts[0])
        b.set_data_device(x.data_ptr(), keepalive=x)
        b.build_device(idx.data_ptr(), dist.data_ptr())
        b.synchronize()
        its.append(b.stats()["n_iters_run"])
        rec.append(bench.recall_at(true10, idx[rows], 10))
        b.close()
        _capi.load_library().nnd_release_pending()
    print(json.dumps({"config": cfg, "knobs": env, "recall_mean": round(float(np.mean(rec)), 5), "recall_std": round(float(np.std(rec)), 5),
                      "recalls": [round(r, 5) for r in rec], "iters": its}), flush=True)
xh = x.cpu().numpy()
rows_h = rows.cpu().numpy()
t10 = true10.cpu().numpy()
rec = []
for s in range(1, o_seeds + 1):
    t0 = time.time()
    oi, _ = O.build_index(xh, metric, n_neighbors=k, n_trees=T, random_state=s, n_threads=128, kind="fast")
    rec.append(O.recall(t10, oi[rows_h]))
    print(json.dumps({"config": cfg, "oracle_seed": s, "recall": round(rec[-1], 5), "seconds": round(time.time() - t0, 1)}), flush=True)

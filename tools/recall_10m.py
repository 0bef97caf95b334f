"""One-off: does the reference ALGORITHM lose recall with n on this generator the way the GPU build
does?  BASELINE configs[3]'s point set (10 M x 128, euclidean, k = 15, 12 trees = the reference default) through the CPU
oracle on the GPU box's host cores and through the GPU build; recall@10 of both on the same sample of rows.
usage: python tools/recall_10m.py [n] [threads]   (writes one JSON line; ~6 minutes of CPU time at n = 1e7)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402  (checker)
from pynndescent_amd import _capi  # noqa: E402
from tools.qbench import sift_like_np  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    k, n_trees = 15, 12
    x = sift_like_np(n, 128, seed=3)
    rng_state, _, ts = O.draw_rng_states(1, n_trees)
    t0 = time.perf_counter()
    b = _capi.Builder(n, 128, 0, k, n_trees, O.default_leaf_size(k), 200, min(60, k), O.default_n_iters(n), 0.001, rng_state, ts[0])
    b.set_data_host(x)
    b.make_forest(); b.init_from_leaves(); b.init_random(); b.descent()
    gidx, _ = b.finalize()
    st = b.stats()
    b.close()
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    oidx, _ = O.build_index(x, "euclidean", n_neighbors=k, n_trees=n_trees, random_state=1, n_threads=threads, kind="fast")
    t_cpu = time.perf_counter() - t0
    rows = np.random.RandomState(5).choice(n, 400, replace=False)
    ti, _ = O.brute_force_knn(x, 10, "euclidean", rows=rows)
    print(json.dumps({"n": n, "k": k, "n_trees": n_trees, "recall_at_10_gpu": round(float(O.recall(ti, gidx[rows])), 4),
                      "recall_at_10_oracle": round(float(O.recall(ti, oidx[rows])), 4), "gpu_iters": st["n_iters_run"],
                      "gpu_wall_s_incl_h2d": round(t_gpu, 2), "oracle_seconds": round(t_cpu, 1), "oracle_threads": threads,
                      "sample_rows": 400}))


if __name__ == "__main__":
    main()

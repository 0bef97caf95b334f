"""prepare() and query() of a GPU-built index at scale (SURVEY.md section 8f rows 2 and 4 to the measurement bar):
build 1 M x 128 (euclidean, k = 15), time prepare() by stage, then batched queries: queries / s and recall@10 vs exact.
usage: python tools/bench_search.py [n] [n_queries]      (no torch; prints one JSON line)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pynndescent_amd import NNDescent  # noqa: E402
from pynndescent_amd.search_tree import make_hub_tree  # noqa: E402
from tools.qbench import sift_like_np  # noqa: E402


def exact(x, q, k):
    out = np.empty((q.shape[0], k), np.int64)
    xn = (x.astype(np.float64) ** 2).sum(1)
    for a in range(0, q.shape[0], 64):
        qq = q[a:a + 64].astype(np.float64)
        dd = (qq * qq).sum(1)[:, None] + xn[None, :] - 2.0 * qq @ x.T.astype(np.float64)
        out[a:a + 64] = np.argpartition(dd, k, axis=1)[:, :k]
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
    allx = sift_like_np(n + nq, 128, seed=1)
    x, q = allx[:n], allx[n:]
    t0 = time.perf_counter()
    index = NNDescent(x, "euclidean", n_neighbors=15, n_trees=8, random_state=3)
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    tree = make_hub_tree(x, index._neighbor_graph[0], "euclidean", 30, 200)
    t_tree = time.perf_counter() - t0
    t0 = time.perf_counter()
    index.prepare()
    t_prepare = time.perf_counter() - t0
    index.query(q[:256], k=10, epsilon=0.1)  # warm
    res = {"n": n, "n_queries": nq, "build_s_incl_h2d_d2h": round(t_build, 3), "hub_tree_alone_s": round(t_tree, 3),
           "hub_tree_nodes": int(tree.children.shape[0]), "prepare_total_s": round(t_prepare, 3), "queries": []}
    rows = np.arange(0, nq, max(1, nq // 500))
    truth = exact(x, q[rows], 10)
    for eps in (0.0, 0.1, 0.2):
        t0 = time.perf_counter()
        qi, qd = index.query(q, k=10, epsilon=eps)
        dt = time.perf_counter() - t0
        rec = float(np.mean([len(np.intersect1d(t, a)) / 10.0 for t, a in zip(truth, qi[rows])]))
        res["queries"].append({"epsilon": eps, "queries_per_s_host_to_host": round(nq / dt, 1), "recall_at_10": round(rec, 4)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()

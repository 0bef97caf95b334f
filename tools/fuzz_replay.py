"""Replay ONE configuration of tools/fuzz_parity.py (same random stream: N, seed, the build seed printed in its log line) over
several build seeds on both sides: is a gap between the GPU build and the oracle systematic or the seed noise of a stuck graph?
usage: fuzz_replay.py N seed build_seed [n_seeds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle as O
from pynndescent_amd import _capi
from tests.util_data import clustered


def find(count, seed, want):
    rs = np.random.RandomState(seed)
    for t in range(count):
        n = int(rs.choice([300, 1200, 2500, 5000]))
        d = int(rs.choice([2, 3, 7, 16, 31, 32, 33, 64, 96, 100, 129, 160, 257]))
        k = int(rs.choice([2, 5, 10, 15, 16, 17, 24, 30, 33, 48, 64]))
        k = min(k, n - 1)
        metric = str(rs.choice(["euclidean", "cosine"]))
        n_trees = int(rs.choice([1, 3, 8, 12]))
        leaf = rs.choice([None, None, 20, 60, 100, 200])
        leaf = None if leaf is None else int(leaf)
        mc = rs.choice([None, None, None, 5, 20, 40, 60])
        mc = None if mc is None else int(mc)
        dseed = int(rs.randint(1 << 30))
        nonneg = (metric == "euclidean" and rs.rand() < 0.5)
        s = int(rs.randint(1 << 30))
        if s == want:
            x = clustered(n, d, max(2, min(d, 8)), 12, seed=dseed, nonneg=nonneg)
            return x, dict(n=n, d=d, k=k, metric=metric, n_trees=n_trees, leaf_size=leaf, max_candidates=mc)
    raise SystemExit("configuration not found")


def gpu(x, cfg, s, join_blocks, own_forest=True, leaves_in=None):
    n, d = x.shape
    k, metric, n_trees = cfg["k"], cfg["metric"], cfg["n_trees"]
    rng_state, _, ts = O.draw_rng_states(s, max(n_trees, 1))
    ls = O.default_leaf_size(k) if cfg["leaf_size"] is None else cfg["leaf_size"]
    emc = min(60, k) if cfg["max_candidates"] is None else cfg["max_candidates"]
    n_iters = O.default_n_iters(n)
    b = _capi.Builder(n, d, O.METRICS[metric], k, n_trees, ls, 200, emc, n_iters, 0.001, rng_state, ts[0], join_blocks=join_blocks)
    b.set_data_host(x)
    b.make_forest()
    leaves = b.leaf_array()
    b.reset_graph()
    b.init_from_leaves()
    b.init_random()
    its = 0
    for _ in range(n_iters):
        its += 1
        if b.descent_iter() <= 0.001 * k * n:
            break
    idx, _ = b.finalize()
    b.close()
    return idx, leaves, rng_state, emc, n_iters, its


def main():
    count, seed, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    n_seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    x, cfg = find(count, seed, want)
    print(cfg, flush=True)
    kt = min(cfg["k"], 10)
    ti, _ = O.brute_force_knn(x, kt, cfg["metric"])
    rows = {"gpu_auto": [], "gpu_one_launch": [], "oracle": []}
    for q in range(n_seeds):
        s = want if q == 0 else want + 7919 * q
        idx, leaves, rng_state, emc, n_iters, its = gpu(x, cfg, s, 0)
        idx1, _, _, _, _, its1 = gpu(x, cfg, s, 1)
        oidx, _ = O.nn_descent(x, cfg["k"], rng_state.copy(), emc, cfg["metric"], n_iters, 0.001, leaves, n_threads=8)
        r = (O.recall(ti, idx), O.recall(ti, idx1), O.recall(ti, oidx))
        rows["gpu_auto"].append(r[0]); rows["gpu_one_launch"].append(r[1]); rows["oracle"].append(r[2])
        print("seed %d: gpu (sub-steps auto, %d iterations) %.4f   gpu (one launch per iteration, %d) %.4f   oracle %.4f  leaves %d" % (
            s, its, r[0], its1, r[1], r[2], leaves.shape[0]), flush=True)
    for name, v in rows.items():
        print("%-16s mean %.4f  sd %.4f  min %.4f  max %.4f" % (name, np.mean(v), np.std(v), np.min(v), np.max(v)))


if __name__ == "__main__":
    main()

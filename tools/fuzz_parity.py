"""Randomised parity sweep over random (n, d, k, metric, n_trees, leaf_size, max_candidates): the GPU build against the CPU
oracle's NN-descent run ON THE GPU'S OWN LEAF ARRAY (same trees: with one tree and a handful of leaves the luck of the
tree -- whether a tiny leaf injects random edges -- moves recall by 0.1-0.3 on clustered data and says nothing about
parity), plus structural invariants.  usage: fuzz_parity.py [N] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle as O
from pynndescent_amd import _capi
from tests.util_data import clustered


def main(count, seed):
    rs = np.random.RandomState(seed)
    bad = 0
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
        x = clustered(n, d, max(2, min(d, 8)), 12, seed=int(rs.randint(1 << 30)), nonneg=(metric == "euclidean" and rs.rand() < 0.5))
        s = int(rs.randint(1 << 30))
        cfg = dict(n=n, d=d, k=k, metric=metric, n_trees=n_trees, leaf_size=leaf, max_candidates=mc, seed=s)
        try:
            rng_state, _, ts = O.draw_rng_states(s, max(n_trees, 1))
            ls = O.default_leaf_size(k) if leaf is None else leaf
            emc = min(60, k) if mc is None else mc
            n_iters = O.default_n_iters(n)
            b = _capi.Builder(n, d, O.METRICS[metric], k, n_trees, ls, 200, emc, n_iters, 0.001, rng_state, ts[0])
            b.set_data_host(x)
            b.make_forest()
            leaves = b.leaf_array()
            b.reset_graph()
            b.init_from_leaves()
            b.init_random()
            for _ in range(n_iters):
                if b.descent_iter() <= 0.001 * k * n:
                    break
            idx, dist = b.finalize()
            b.close()
            oidx, _ = O.nn_descent(x, k, rng_state.copy(), emc, metric, n_iters, 0.001, leaves, n_threads=8)
            kt = min(k, 10)
            ti, _ = O.brute_force_knn(x, kt, metric)
            rg, ro = O.recall(ti, idx), O.recall(ti, oidx)
            # one tree or a few hundred points: the random fill (different RNG on the two sides) decides how many escape
            # routes a stuck graph gets; recall then moves by several percent from run to run on BOTH sides
            tol = 0.08 if (n_trees == 1 or n < 1000) else 0.03
            ok = abs(rg - ro) <= tol
            if not ok:
                # a stuck graph's recall has a standard deviation of 0.03-0.04 over build seeds on EITHER side (tools/fuzz_replay.py,
                # profiles/r06_fuzz_replay60.log: one 0.93 vs 0.81 pair, means over 60 seeds 0.9346 vs 0.9304): a single pair beyond
                # the tolerance is re-drawn 16 times and the MEANS are compared
                from tools.fuzz_replay import gpu as gpu_build
                gs, os_ = [], []
                for q in range(16):
                    s2 = s + 7919 * (q + 1)
                    gi, lv, rst, emc2, nit2, _ = gpu_build(x, cfg, s2, 0)
                    oi2, _ = O.nn_descent(x, k, rst.copy(), emc2, metric, nit2, 0.001, lv, n_threads=8)
                    gs.append(O.recall(ti, gi)); os_.append(O.recall(ti, oi2))
                ok = abs(float(np.mean(gs)) - float(np.mean(os_))) <= 0.03
                print("     re-drawn over 16 seeds: gpu mean %.4f (sd %.4f)  oracle mean %.4f (sd %.4f)" % (
                    np.mean(gs), np.std(gs), np.mean(os_), np.std(os_)))
            filled = idx >= 0
            big = np.where(filled, dist, np.inf)
            ok &= bool(np.all(big[:, 1:] >= big[:, :-1]))
            ok &= all(len(set(r[f].tolist())) == int(f.sum()) for r, f in zip(idx[::37], filled[::37]))
            print("%s recall gpu %.4f oracle %.4f %s" % ("ok  " if ok else "FAIL", rg, ro, cfg))
            bad += 0 if ok else 1
        except Exception as e:  # noqa: BLE001
            print("EXC ", type(e).__name__, str(e)[:120], cfg)
            bad += 1
        sys.stdout.flush()
    print("failures:", bad, "of", count)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)

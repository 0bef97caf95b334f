"""Tiny inputs through the class API: n = 1 .. 40 against every k <= n + 2 (structural invariants; exact neighbours expected when
the build has enough room to see every pair).  usage: tiny_sizes.py"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pynndescent_amd import NNDescent


def run(verbose=True):
    bad = 0
    n_ok = 0
    rs = np.random.RandomState(3)
    for n in (1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 33, 40):
        for d in (1, 3, 32):
            x = rs.standard_normal((n, d)).astype(np.float32)
            for metric in ("euclidean", "cosine"):
                for k in sorted(set([1, 2, 3, max(1, n - 1), n, n + 2])):
                    try:
                        with warnings.catch_warnings():
                            warnings.simplefilter("ignore")
                            idx, dist = NNDescent(x, metric, n_neighbors=k, random_state=1).neighbor_graph
                        ok = idx.shape == (n, k)
                        f = idx >= 0
                        ok &= bool(idx.max() < n)
                        ok &= bool(np.all(np.isfinite(dist[f])))
                        big = np.where(f, dist, np.inf)
                        ok &= bool(np.all(big[:, 1:] >= big[:, :-1]))
                        ok &= all(len(set(r[m].tolist())) == int(m.sum()) for r, m in zip(idx, f))
                        ok &= bool(np.all(f.sum(1) == min(k, n)))  # every row holds min(k, n) distinct points (itself included)
                        msg = "ok  " if ok else "FAIL"
                    except (ValueError, NotImplementedError) as e:
                        ok, msg = True, "rej "  # refused up front with a message
                        idx = None
                        print(msg, dict(n=n, d=d, k=k, metric=metric), type(e).__name__, str(e)[:90], flush=True)
                        continue
                    except Exception as e:  # noqa: BLE001
                        ok, msg = False, "EXC "
                        print(msg, dict(n=n, d=d, k=k, metric=metric), type(e).__name__, str(e)[:120], flush=True)
                        bad += 1
                        continue
                    n_ok += 1 if ok else 0
                    if not ok:
                        bad += 1
                        print(msg, dict(n=n, d=d, k=k, metric=metric), "filled per row", f.sum(1)[:8], flush=True)
    if verbose:
        print("builds ok:", n_ok, "failures:", bad)
    return n_ok, bad


if __name__ == "__main__":
    sys.exit(1 if run()[1] else 0)

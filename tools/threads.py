"""Independent indexes built from several Python threads at once on one device (ctypes releases the GIL): every result must equal the
one the same call gives alone.  usage: threads.py [threads] [rounds]"""
import os
import sys
import threading
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pynndescent_amd import NNDescent


def job(q):
    rs = np.random.RandomState(100 + q)
    n = int(rs.choice([3000, 20000, 150000, 260000]))
    d = int(rs.choice([16, 64, 100]))
    k = int(rs.choice([10, 15, 30]))
    metric = "euclidean" if q % 2 == 0 else "cosine"
    x = rs.standard_normal((n, d)).astype(np.float32)
    return x, metric, k, q


def build(x, metric, k, seed):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        idx, dist = NNDescent(x, metric, n_neighbors=k, random_state=seed).neighbor_graph
    return idx.copy(), dist.copy()


def run(n_threads=4, rounds=3, verbose=True):
    jobs = [job(q) for q in range(n_threads * rounds)]
    alone = [build(*j) for j in jobs]
    got = [None] * len(jobs)
    errs = []

    def worker(t):
        try:
            for r in range(rounds):
                q = t * rounds + r
                got[q] = build(*jobs[q])
        except Exception as e:  # noqa: BLE001
            errs.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    bad = len(errs)
    for q, (a, g) in enumerate(zip(alone, got)):
        same = g is not None and np.array_equal(a[0], g[0]) and np.array_equal(a[1], g[1])
        if not same:
            bad += 1
            if verbose:
                n_diff = -1 if g is None else int((a[0] != g[0]).any(1).sum())
                print("job %d (n=%d d=%d k=%d %s): differs from the build made alone (%d rows)" % (
                    q, jobs[q][0].shape[0], jobs[q][0].shape[1], jobs[q][2], jobs[q][1], n_diff))
    if verbose:
        for e in errs:
            print("thread %d: %s" % e)
        print("%d threads x %d builds: %d mismatches / errors" % (n_threads, rounds, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 3) else 0)

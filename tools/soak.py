"""Soak: many builds in one process (class API, two geometries alternating, plus update() and prepare()/query() now and then);
device memory in use and host RSS must stay flat.  usage: soak.py [builds]"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
import torch

from pynndescent_amd import NNDescent

builds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(0)
xa = rs.standard_normal((400_000, 64)).astype(np.float32)
xb = rs.standard_normal((150_000, 128)).astype(np.float32)
proc = psutil.Process()
t0 = time.time()
log = []
for i in range(builds):
    x, k, metric = (xa, 15, "euclidean") if i % 2 == 0 else (xb, 30, "cosine")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ix = NNDescent(x, metric, n_neighbors=k, random_state=i)
        idx, dist = ix.neighbor_graph
        if i % 25 == 3:
            ix.update(xs_fresh=rs.standard_normal((2000, x.shape[1])).astype(np.float32))
            idx, dist = ix.neighbor_graph
        if i % 40 == 7:
            ix.prepare()
            ix.query(x[:1000], k=10)
    assert idx.shape[1] == k and np.isfinite(dist[idx >= 0]).all()
    del ix, idx, dist
    if i % 20 == 19 or i == builds - 1:
        free, total = torch.cuda.mem_get_info()
        log.append((i + 1, (total - free) / 2**20, proc.memory_info().rss / 2**20))
        print("after %4d builds: device memory in use %8.0f MB   host RSS %8.0f MB   %.0f s" % (log[-1] + (time.time() - t0,)), flush=True)
d0, d1 = log[1][1], log[-1][1]
h0, h1 = log[1][2], log[-1][2]
print("device growth %.0f MB, host growth %.0f MB between build %d and %d" % (d1 - d0, h1 - h0, log[1][0], log[-1][0]))
sys.exit(1 if (d1 - d0 > 256 or h1 - h0 > 512) else 0)

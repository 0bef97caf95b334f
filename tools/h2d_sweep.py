import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pynndescent_amd import _capi
n, d = 1_000_000, 128
x = np.random.RandomState(0).standard_normal((n, d)).astype(np.float32)
for thr in (4, 8, 12, 16):
    os.environ["NND_H2D_THREADS"] = str(thr)
    b = _capi.Builder(n, d, 0, 15, 8, 75, 200, 15, 20, 0.001, [1, 2, 3], [4, 5, 6])
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); b.set_data_host(x); b.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    b.close(); _capi.load_library().nnd_release_pending()
    print("threads", thr, "set_data_host (H2D 488 MB + prep) ms:", [round(t, 2) for t in ts], flush=True)

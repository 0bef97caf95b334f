"""Where the host-inclusive build time goes: nnd_create / nnd_set_data_host / build / finalize_host / destroy, timed apart."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import sift_like
from pynndescent_amd import _capi

n, d, k, T = 1_000_000, 128, 15, 8
x = sift_like(n, d, seed=1, device=torch.device("cuda", 0), sample_seed=100).cpu().numpy()
rs = np.random.RandomState(1234)
lim = np.iinfo(np.int32)
rng = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64)
ts = rs.randint(lim.min + 1, lim.max - 1, size=(T, 3)).astype(np.int64)
for rep in range(3):
    t0 = time.perf_counter()
    b = _capi.Builder(n, d, 0, k, T, 75, 200, 15, 20, 0.001, rng, ts[0])
    t1 = time.perf_counter()
    b.set_data_host(x)
    b.synchronize()
    t2 = time.perf_counter()
    b.make_forest(); b.init_from_leaves(); b.init_random(); b.descent()
    b.synchronize()
    t3 = time.perf_counter()
    idx, dist = b.finalize()
    t4 = time.perf_counter()
    b.close()
    t5 = time.perf_counter()
    print("rep %d: create %.2f ms, set_data_host (H2D %d MB + prep) %.2f ms, build %.2f ms, finalize_host (D2H) %.2f ms, destroy %.2f ms, total %.2f ms"
          % (rep, (t1 - t0) * 1e3, x.nbytes >> 20, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t0) * 1e3))
# plain copies for reference
xt = torch.from_numpy(x)
for name, src in (("pageable", xt), ("pinned", xt.pin_memory())):
    torch.cuda.synchronize(); t0 = time.perf_counter(); y = src.to("cuda:0", non_blocking=False); torch.cuda.synchronize()
    print("H2D %s 488 MB: %.2f ms" % (name, (time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); p = xt.pin_memory(); print("pin_memory (alloc + copy) %.2f ms" % ((time.perf_counter() - t0) * 1e3))
import sklearn.utils
t0 = time.perf_counter(); sklearn.utils.check_array(x, dtype=np.float32, order="C"); print("check_array %.2f ms" % ((time.perf_counter() - t0) * 1e3))

"""One build under NND_POISON (KNOBS library): which kernel reads memory nobody initialised?
usage: NND_POISON=165 PYNND_AMD_LIB=.../lib_kn.so AMD_SERIALIZE_KERNEL=3 AMD_LOG_LEVEL=3 python tools/dbg_poison.py <flags> [n] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from oracle import oracle as O
from pynndescent_amd import _capi
from tools.bench_configs import gen, exact_top10

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 12
d, k, metric = 100, 15, "cosine"
dev = torch.device("cuda", 0)
x = gen(n, d, 24, 2, dev, False)
rows = torch.from_numpy(np.random.RandomState(0).choice(n, 2000, replace=False)).to(dev)
true10 = exact_top10(x, rows, metric)
idx = torch.empty((n, k), dtype=torch.int32, device=dev)
dist = torch.empty((n, k), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
print("MARK build starts", flush=True)
sys.stderr.write("MARK build starts\n"); sys.stderr.flush()
rng_state, _, ts = O.draw_rng_states(1, T)
b = _capi.Builder(n, d, O.METRICS[metric], k, T, O.default_leaf_size(k), 200, min(60, k), O.default_n_iters(n), 0.001, rng_state, ts[0], flags=flags)
b.set_data_device(x.data_ptr(), keepalive=x)
b.build_device(idx.data_ptr(), dist.data_ptr())
b.synchronize()
print("recall", bench.recall_at(true10, idx[rows], 10), b.stats()["updates"][:5], flush=True)

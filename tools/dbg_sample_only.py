"""Timing experiments: ONE first sampling pass at C2' size (nothing consumes the candidates: safe with broken variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from pynndescent_amd import _capi
from tools.bench_configs import CONFIGS, gen
n, d, latent, seed, metric, k, T, nonneg = CONFIGS["c2"]
x = gen(n, d, latent, seed, torch.device("cuda", 0), nonneg)
torch.cuda.synchronize()
rng_state, _, ts = O.draw_rng_states(1, T)
b = _capi.Builder(n, d, O.METRICS[metric], k, T, O.default_leaf_size(k), 200, min(60, k), O.default_n_iters(n), 0.001, rng_state, ts[0])
b.set_data_device(x.data_ptr(), keepalive=x)
b.make_forest(); b.init_from_leaves(); b.init_random()
for _ in range(3):
    b.sample_candidates()
b.synchronize()
print("done", os.path.basename(_capi.LIB_PATH))

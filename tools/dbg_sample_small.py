"""Debug: candidate sampling against the host model (tests/test_gpu_kernels.py), with details of the mismatches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from tests.gpu_util import make_builder
from tests.util_data import clustered
from tests.test_gpu_kernels import _expected_candidates

n, k, mc = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
trees = int(sys.argv[5]) if len(sys.argv) > 5 else 3
x = clustered(n, 24, 6, 40, seed=n % 89)
rng_state, _, _ = O.draw_rng_states(1, 3)
b = make_builder(x, "euclidean", k=k, n_trees=trees, mc=mc, seed=1, flags=flags)
if trees:
    b.make_forest()
    b.init_from_leaves()
b.init_random()
rcap = 32 if mc <= 32 else 64
for it in range(3):
    idx0, _, fl0 = b.graph()
    b.sample_candidates()
    new, old = b.candidates()
    idx1, _, fl1 = b.graph()
    wide = it == 0 and rcap == 32 and k <= 64
    e_new, e_old, x_new, x_old = _expected_candidates(idx0, fl0, rng_state, it, mc, 2 * rcap if wide else rcap, rcap)
    act = new[:, 0] >= 0
    e_act = e_new[:, 0] >= 0
    bad_act = np.nonzero(act != e_act)[0]
    rows = np.nonzero(x_new & act & e_act)[0]
    bad_new = rows[(new[rows] != e_new[rows]).any(1)]
    rows = np.nonzero(x_old & act & e_act)[0]
    bad_old = rows[(old[rows] != e_old[rows]).any(1)]
    print("it", it, "n_act", int(act.sum()), "expected", int(e_act.sum()), "act mismatches", len(bad_act), bad_act[:24].tolist(),
          "| new-list mismatches", len(bad_new), bad_new[:8].tolist(), "| old-list mismatches", len(bad_old), bad_old[:8].tolist(),
          "| id range ok", bool(((new >= -1) & (new < n)).all() and ((old >= -1) & (old < n)).all()), flush=True)
    for v in list(bad_new[:3]):
        print("   v", v, "got", new[v].tolist(), "want", e_new[v].tolist())
    if len(bad_act) or len(bad_new) or len(bad_old):
        break
    b.descent_iter()
b.close()

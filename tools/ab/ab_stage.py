"""A/B helper: time the build stages for one library build (PYNND_AMD_LIB selects the .so) on the bench workload.
usage: PYNND_AMD_LIB=... python tools/ab/ab_stage.py [n_trees ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pynndescent_amd import _capi

def run(n_trees, n=1_000_000, reps=3):
    xd = bench.sift_like(n, 128, seed=1, device="cuda:0", sample_seed=100)
    b = _capi.Builder(n=n, dim=128, metric=0, n_neighbors=15, n_trees=n_trees, leaf_size=75, max_depth=200,
                      max_candidates=15, n_iters=5, delta=0.001, rng_state=(1, 2, 3), tree_rng=(4, 5, 6), device=0)
    torch.cuda.synchronize()  # xd must be complete: the library reads it on its own stream
    b.set_data_device(xd.data_ptr())
    out = []
    for r in range(reps):
        idx = torch.empty((n, 15), dtype=torch.int32, device="cuda")
        dist = torch.empty((n, 15), dtype=torch.float32, device="cuda")
        b.build_device(idx.data_ptr(), dist.data_ptr())
        out.append(b.stats())
    return out[-1]

if __name__ == "__main__":
    trees = [int(a) for a in sys.argv[1:]] or [8]
    for t in trees:
        st = run(t)
        print(t, json.dumps({k: (round(v, 3) if isinstance(v, float) else [round(u, 3) for u in v]) for k, v in st.items()
                             if k.startswith("ms_")}))

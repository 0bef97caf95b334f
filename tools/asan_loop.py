"""Heap-check loop for the host side of comm.hip / shard.hip / capi.hip: many multi-rank builds in ONE process (LOCAL
transport, thread-ranks sharing the GPU), meant to run against the ASan variant of the library:
    LD_PRELOAD=<libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 \
    PYNND_AMD_LIB=pynndescent_amd/_exp/lib_asan.so python tools/asan_loop.py [reps]
(tools/gpu_asan.sh builds the variant and runs this).  Every 8th repetition a rank fails on purpose (the abort path),
every 5th goes through nnd_build_multi (NNDescent(n_devices=2))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pynndescent_amd import NNDescent, _capi
from tests.test_gpu_sharded import _run_local
from tests.util_data import clustered

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
x_big = torch.from_numpy(clustered(200_000, 32, 8, 400, seed=3)).cuda()   # forest sharded by cell
x_small = torch.from_numpy(clustered(6000, 32, 8, 40, seed=31)).cuda()    # forest split by tree
x_host = clustered(20_000, 32, 8, 100, seed=7)
t0 = time.time()
n_fail = 0
for i in range(reps):
    world = (8, 3, 2, 5)[i % 4]
    if i % 8 == 7:
        try:
            _run_local(x_big, world, "euclidean", 15, n_trees=4, seed=i, flags=_capi.NND_FLAG_TEST_FAIL)
            raise SystemExit("repetition %d: the failing rank went unnoticed" % i)
        except AssertionError:
            n_fail += 1
    elif i % 5 == 4:
        g = NNDescent(x_host, "euclidean", n_neighbors=15, n_trees=4, random_state=i, n_devices=2, devices=[0, 0])._neighbor_graph[0]
        assert g.shape == (20_000, 15) and (g >= 0).all()
    else:
        x = x_big if i % 2 == 0 else x_small
        idx, _, infos = _run_local(x, world, "euclidean", 15, n_trees=4, seed=i)
        assert int(idx.min()) >= 0 and idx.shape[0] == x.shape[0]
    if i % 20 == 19:
        print("repetition %d, %.0f s, %d deliberate failures handled" % (i + 1, time.time() - t0, n_fail), flush=True)
print("done: %d repetitions, %d deliberate failures, no heap error reported" % (reps, n_fail))

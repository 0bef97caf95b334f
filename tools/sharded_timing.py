"""Time the row-sharded build flow with G ranks as THREADS on one GPU (ThreadComm): what the host flow + the extra
export / import / proposal kernels cost relative to the plain single-handle build of the same point set.
(The ranks share one GPU, so this is an upper bound of the per-rank overhead, not a scaling number.)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pynndescent_amd import sharded


def run(world, n_total):
    dev = torch.device("cuda", 0)
    x = bench.sift_like(n_total, 128, seed=1, device=dev, sample_seed=100)
    ranges = sharded.shard_ranges(n_total, world)
    comms = sharded.ThreadComm.make(world)
    times = [None] * world
    stats = [None] * world

    def work(r):
        torch.cuda.set_device(0)
        lo, hi = ranges[r]
        sb = sharded.ShardedBuilder(comms[r], [b - a for a, b in ranges], 128, "euclidean", 15, 8, seed=1234, device_index=0)
        xl = x[lo:hi].contiguous()
        for rep in range(3):
            comms[r].barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, info = sb.build(xl)
            torch.cuda.synchronize()
            comms[r].barrier()
            times[r] = (time.perf_counter() - t0) * 1e3
        stats[r] = info["stats"]
        sb.close()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    st = stats[0]
    print("world %d n_total %d: %.1f ms per build (rank 0 stages: prep %.1f forest %.1f leaf %.1f join %.1f sample %.1f merge %.1f finalize %.1f)" % (
        world, n_total, max(times), st["ms_prep"], st["ms_forest"], st["ms_leaf_init"], sum(st["ms_join"]), sum(st["ms_sample"]),
        sum(st["ms_merge"]), st["ms_finalize"]))
    sys.stdout.flush()


if __name__ == "__main__":
    for w in (1, 2, 4):
        run(w, 1_000_000)

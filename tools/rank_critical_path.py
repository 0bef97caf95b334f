"""Per-rank critical path of the row-sharded build, measured on ONE GPU.

G ranks run as threads of this process on the same GPU (LOCAL transport) in SERIAL mode: between two exchanges only one
rank's compute section runs at a time, with the GPU to itself, and is timed (nnd_shard_info.section_ms).  On G real GPUs
the sections of different ranks run concurrently, so the build's critical path is

    sum over sections of max over ranks (section time)  +  the exchanges,

the exchanges priced from the payload bytes every rank sent (nnd_shard_info.section_bytes) at an xGMI rate: 7 links per
GPU, --link-gbs effective GB/s each (default 45: what RCCL send/recv groups reach on a ~64 GB/s link), plus --latency-us
per exchange.  Prints one JSON line.

    python tools/rank_critical_path.py --world 8 --n 10000000 --trees 12
"""
import argparse
import json
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exact_knn_sample, recall_at, sift_like  # noqa: E402
from pynndescent_amd import _capi, sharded  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--trees", type=int, default=12)
    ap.add_argument("--link-gbs", type=float, default=45.0)
    ap.add_argument("--latency-us", type=float, default=40.0)
    ap.add_argument("--builds", type=int, default=2, help="the last build is reported (the first one pays allocations)")
    ap.add_argument("--by-tree", action="store_true", help="forest split by tree (the round-3 scheme) instead of sharded by cell")
    ap.add_argument("--one-gpu", action="store_true", help="also time the same set on one GPU (plain builder): the speed-up's numerator")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    G, n = args.world, args.n
    x = sift_like(n, args.dim, seed=1, device=dev, sample_seed=7)
    torch.cuda.synchronize()
    ranges = sharded.shard_ranges(n, G)
    sizes = [b - a for a, b in ranges]
    grp = sharded.LocalGroup(G)
    grp[0].set_serial(True)
    infos, outs, err = [None] * G, [None] * G, []

    def run(r):
        sb = None
        try:
            torch.cuda.set_device(0)
            sb = sharded.ShardedBuilder(grp[r], sizes, args.dim, "euclidean", args.k, args.trees, seed=9, device_index=0,
                                        flags=_capi.NND_FLAG_TEST_FOREST_BY_TREE if args.by_tree else 0)
            lo, hi = ranges[r]
            xl = x[lo:hi].contiguous()
            for _ in range(args.builds):
                idx, dist, info = sb.build(xl)
            infos[r] = info
            outs[r] = idx.clone()
        except Exception as e:
            err.append("rank %d: %r" % (r, e))
            _capi.load_library().nnd_comm_abort(grp[r]._h)
        finally:
            if sb is not None:
                sb.close()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(G)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    grp.close()
    if err:
        sys.exit("\n".join(err))
    ns = min(len(i["section_ms"]) for i in infos)
    sec = np.array([i["section_ms"][:ns] for i in infos])      # (G, sections)
    byt = np.array([i["section_bytes"][:ns] for i in infos])   # payload sent after each section
    compute_cp = float(sec.max(0).sum())
    egress = 7.0 * args.link_gbs * 1e9
    n_exch = int((byt.max(0) > 0).sum())
    exch_ms = float((byt.max(0) / egress).sum() * 1e3 + n_exch * args.latency_us * 1e-3)
    # round 6: the threshold / neighbour-id gather of every iteration runs on the second channel beside the offer exchange and the
    # second half of the sampling (section gather_section[i], timed apart in serial mode); what it exceeds them by is exposed
    gth = np.array([i.get("gather_bytes", []) + [0] * 64 for i in infos])[:, :64].max(0)
    gsec = infos[0].get("gather_section", [])
    gather_inline_ms = gather_exposed_ms = 0.0
    for i, gs in enumerate(gsec):
        if gth[i] <= 0:
            continue
        g_ms = gth[i] / egress * 1e3 + args.latency_us * 1e-3
        gather_inline_ms += g_ms
        if gs >= 1 and gs < ns:
            # beside the SECTION only: the offer exchange in front of it runs at the same time too, but over the same links -- the
            # bytes of both have to pass, so no credit is taken for that part of the overlap
            gather_exposed_ms += max(0.0, g_ms - float(sec.max(0)[gs]))
        else:
            gather_exposed_ms += g_ms
    exch_ms += gather_exposed_ms
    # the point-set all-gather is not attached to a section.  Forest by cell: it runs on the second channel beside the first
    # n_sections_overlap sections (own rows + sample only) and the exchanges between them; what it exceeds them by is exposed
    allgather_bytes = (n - max(sizes)) * args.dim * 4  # received per rank; ring / direct: each link carries 1/7 of it
    allgather_full_ms = allgather_bytes / egress * 1e3
    nov = min(i.get("n_sections_overlap", 0) for i in infos)
    beside_ms = float(sec.max(0)[:nov].sum()) if nov else 0.0
    allgather_ms = max(0.0, allgather_full_ms - beside_ms)
    one_gpu_ms = None
    if args.one_gpu:
        import time
        lim = np.iinfo(np.int32)
        rs = np.random.RandomState(9)
        rng_state = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64)
        _ = rs.randint(lim.min + 1, lim.max - 1, 3)
        tstate = rs.randint(lim.min + 1, lim.max - 1, size=(args.trees, 3)).astype(np.int64)
        b1 = _capi.Builder(n, args.dim, 0, args.k, args.trees, max(60, min(256, 5 * args.k)), 200, min(60, args.k), max(5, int(round(np.log2(n)))),
                           0.001, rng_state, tstate[0], device=0)
        oi = torch.empty((n, args.k), dtype=torch.int32, device=dev)
        od = torch.empty((n, args.k), dtype=torch.float32, device=dev)
        b1.set_data_device(x.data_ptr(), keepalive=x)
        for _ in range(2):
            b1.synchronize()
            t1 = time.perf_counter()
            b1.build_device(oi.data_ptr(), od.data_ptr())
            b1.synchronize()
            one_gpu_ms = (time.perf_counter() - t1) * 1e3
        b1.close()
        del oi, od
    idx = torch.cat(outs)
    rows = torch.from_numpy(np.random.RandomState(0).choice(n, 2000, replace=False)).to(dev)
    rec = recall_at(exact_knn_sample(x, rows, 10), idx[rows], 10)
    st0 = infos[0]["stats"]
    out = {
        "what": "per-rank critical path of the sharded build, thread-ranks in serial mode on one MI355X",
        "world": G, "n": n, "dim": args.dim, "k": args.k, "n_trees": args.trees, "iters": infos[0]["iters"],
        "recall_at_10": round(rec, 4),
        "compute_critical_path_ms": round(compute_cp, 2),
        "compute_sum_all_ranks_ms": round(float(sec.sum()), 2),
        "per_rank_compute_ms": [round(float(v), 2) for v in sec.sum(1)],
        "sections_max_ms": [round(float(v), 2) for v in sec.max(0)],
        "sections_min_ms": [round(float(v), 2) for v in sec.min(0)],
        "exchange_bytes_max_per_rank": [int(v) for v in byt.max(0)],
        "modelled_exchange_ms": round(exch_ms, 2),
        "gather_ms_if_inline": round(gather_inline_ms, 2), "gather_exposed_ms": round(gather_exposed_ms, 2),
        "modelled_allgather_ms": round(allgather_full_ms, 2),
        "allgather_exposed_ms": round(allgather_ms, 2),
        "sections_beside_the_allgather": nov,
        "forest_by_cell": bool(infos[0].get("forest_by_cell")),
        "forest_positions_per_rank": [i.get("forest_positions", 0) for i in infos],
        "section0_max_over_min": round(float(sec.sum(1).max() / max(sec.sum(1).min(), 1e-9)), 3),
        "one_gpu_same_set_ms": None if one_gpu_ms is None else round(one_gpu_ms, 2),
        "speedup_modelled": None if one_gpu_ms is None else round(one_gpu_ms / (compute_cp + exch_ms + allgather_ms), 2),
        "model": "7 xGMI links x %.0f GB/s effective per GPU, %.0f us per exchange" % (args.link_gbs, args.latency_us),
        "critical_path_ms": round(compute_cp + exch_ms + allgather_ms, 2),
        "rank0_stage_ms": {"prep": round(st0["ms_prep"], 2), "forest": round(st0["ms_forest"], 2), "leaf_init": round(st0["ms_leaf_init"], 2),
                           "join": round(sum(st0["ms_join"]), 2), "sample": round(sum(st0["ms_sample"]), 2),
                           "merge": round(sum(st0["ms_merge"]), 2), "finalize": round(st0["ms_finalize"], 2)},
        "deferred_proposals": [sum(i["deferred"]) for i in infos],
        "local_trees": [i["local_trees"] for i in infos],
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()

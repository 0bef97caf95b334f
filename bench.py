"""bench.py -- index-build throughput of the MI355X NN-Descent builder (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete index build (prep -> RP forest -> leaf seeding -> random fill ->
NN-descent to the reference's stop rule -> exact-distance finalize) over a synthetic point set that
is already resident in HBM, with the neighbour graph left resident in HBM.  Workload at N=1:
BASELINE.json configs[1] -- "SIFT-1M (1e6 x 128 float32) euclidean k=15, RP-tree init n_trees=8" --
on the seeded SIFT-like stand-in of SURVEY.md section 8d (no dataset files / network here).

Rank 0 prints ONE JSON line (contract in the task statement) carrying, besides the metric:
  roofline     : dominant kernel's algorithmic HBM bytes / its HIP-event duration vs 8 TB/s
  cpu_baseline : the CPU oracle (restatement of the reference algorithm, oracle/) timed on this
                 box's host cores on a bounded sample of the same workload
  recall_at_10 : recall vs exact brute force on a sample of points (reference-test convention)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402  (device memory, streams, torch.distributed: plumbing only)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def sift_like(n, d, seed, device, latent=16, n_clusters=1024, noise=0.3):
    """SURVEY.md section 8d C2': 1024-component Gaussian mixture in a 16-dim latent, random linear map to
    d dims, + noise, shifted non-negative, scaled to ~[0, 218]; generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    centres = torch.randn(n_clusters, latent, generator=g, device=device) * 3.0
    assign = torch.randint(0, n_clusters, (n,), generator=g, device=device)
    z = centres[assign] + torch.randn(n, latent, generator=g, device=device)
    proj = torch.randn(latent, d, generator=g, device=device) / (latent ** 0.5)
    x = z @ proj + noise * torch.randn(n, d, generator=g, device=device)
    x = x - x.min()
    x = x * (218.0 / x.max())
    return x.contiguous().float()


def exact_knn_sample(x, rows, k):
    """Brute-force ground truth (self included) for a sample of rows, float64 refinement of the top 4k."""
    q = x[rows]
    d2 = (q * q).sum(1, keepdim=True) + (x * x).sum(1)[None, :] - 2.0 * (q @ x.T)
    cand = d2.topk(4 * k, dim=1, largest=False).indices
    qq = q.double()[:, None, :]
    dd = ((qq - x[cand].double()) ** 2).sum(-1)
    order = dd.argsort(dim=1)[:, :k]
    return torch.gather(cand, 1, order)


def recall_at(true_idx, approx_idx, k_true=10, cols=None):
    t = true_idx[:, :k_true].cpu().numpy()
    a = approx_idx.cpu().numpy() if cols is None else approx_idx[:, :cols].cpu().numpy()
    hits = sum(np.isin(tr, ar).sum() for tr, ar in zip(t, a))
    return hits / float(t.shape[0] * k_true)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--n-trees", type=int, default=8)
    ap.add_argument("--join-blocks", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from oracle import oracle as O  # cpu_baseline leg + defaults only
    from pynndescent_amd import _capi

    n, d, k = args.n, args.dim, args.k
    x = sift_like(n, d, seed=1 + rank, device=device)
    torch.cuda.synchronize()
    out_idx = torch.empty((n, k), dtype=torch.int32, device=device)
    out_dist = torch.empty((n, k), dtype=torch.float32, device=device)
    rng_state, _, tree_states = O.draw_rng_states(1234 + rank, args.n_trees)
    n_iters = O.default_n_iters(n)
    builder = _capi.Builder(n, d, _capi.NND_METRIC_SQEUCLIDEAN, k, args.n_trees, O.default_leaf_size(k), 200,
                            min(60, k), n_iters, 0.001, rng_state, tree_states[0], device=local_rank,
                            join_blocks=args.join_blocks)

    def step():
        builder.set_data_device(x.data_ptr(), keepalive=x)
        builder.build_device(out_idx.data_ptr(), out_dist.data_ptr())

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        builder.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stage = {"forest": 0.0, "leaf_init": 0.0, "descent": 0.0, "join": 0.0, "sample": 0.0, "merge": 0.0,
             "finalize": 0.0, "prep": 0.0, "random_init": 0.0}
    join_bytes = join_ms = 0.0
    leaf_bytes = 0.0
    n_join_launches = 0
    last = None
    for _ in range(args.steps):
        step()
        st = builder.stats()  # per-stage HIP-event timings taken on the library's own stream
        last = st
        stage["forest"] += st["ms_forest"]
        stage["leaf_init"] += st["ms_leaf_init"]
        stage["descent"] += st["ms_descent"]
        stage["finalize"] += st["ms_finalize"]
        stage["prep"] += st["ms_prep"]
        stage["random_init"] += st["ms_random_init"]
        stage["join"] += sum(st["ms_join"])
        stage["sample"] += sum(st["ms_sample"])
        stage["merge"] += sum(st["ms_merge"])
        join_ms += sum(st["ms_join"])
        join_bytes += sum(st["join_rows"]) * 4.0 * builder_dp(d)
        leaf_bytes += st["leaf_rows"] * 4.0 * builder_dp(d)
        n_join_launches += st["n_iters_run"] * args.join_blocks
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1000.0 / args.steps
    value = world * n * args.steps / elapsed

    result = None
    if rank == 0:
        # recall vs exact brute force on a sample (outside the timed region)
        rs = np.random.RandomState(0)
        rows = torch.from_numpy(rs.choice(n, size=min(2000, n), replace=False)).to(device)
        true_idx = exact_knn_sample(x, rows, 10)
        rec_all = recall_at(true_idx, out_idx[rows], 10)
        rec_strict = recall_at(true_idx, out_idx[rows], 10, cols=10)
        # distances: returned (alt-space) vs float64 truth for the returned pairs
        nb = x[out_idx[rows].long()].double()
        truth = ((x[rows].double()[:, None, :] - nb) ** 2).sum(-1)
        rel = ((out_dist[rows].double() - truth).abs() / truth.clamp_min(1e-30))[truth > 0].max().item()

        # roofline for the dominant kernel (by measured time): the local join gathers C_i candidate rows
        # of dp*4 bytes each per launch (SURVEY.md section 8d B_iter, dominant term)
        dominant = max(("join", "leaf_init", "forest"), key=lambda s: stage[s])
        steps = float(args.steps)
        if dominant == "join":
            achieved = join_bytes / (join_ms * 1e-3) / 1e9 if join_ms > 0 else 0.0
            kernel = "k_local_join"
        elif dominant == "leaf_init":
            achieved = leaf_bytes / (stage["leaf_init"] * 1e-3) / 1e9
            kernel = "k_leaf_join"
        else:
            tree_bytes = steps * (n * 4.0 * builder_dp(d) * last["tree_levels"] + n * 8.0 * last["tree_levels"] * args.n_trees)
            achieved = tree_bytes / (stage["forest"] * 1e-3) / 1e9
            kernel = "rp_forest (k_margin et al.)"
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "join_GBps": round(join_bytes / (join_ms * 1e-3) / 1e9, 2) if join_ms > 0 else None,
                    "join_avg_launch_ms": round(join_ms / max(n_join_launches, 1), 4)}

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            ns = min(args.cpu_sample, n)
            xs = x[:ns].cpu().numpy()
            O.build()
            t1 = time.perf_counter()
            O.build_index(xs, "euclidean", n_neighbors=k, n_trees=args.n_trees, random_state=1234,
                          n_threads=cores, kind="fast")
            dt = time.perf_counter() - t1
            cpu = {"value": round(ns / dt, 1), "unit": "points/s", "cores": cores, "kind": "port",
                   "sample": "first %d points of the same synthetic set, same k/n_trees/defaults, CPU restatement of "
                             "the reference algorithm (numba unavailable), -O3 -ffast-math + OpenMP" % ns}

        result = {
            "metric": "index build: points indexed/sec (recall@10 vs brute force reported alongside)",
            "value": round(value, 1),
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "SIFT-like %dx%d float32 euclidean k=%d n_trees=%d (BASELINE configs[1] stand-in, "
                                   "SURVEY 8d C2')" % (n, d, k, args.n_trees),
                       "parallelism": "1 GPU" if world == 1 else "%d independent row shards of %d points" % (world, n),
                       "join_blocks": args.join_blocks},
            "recall_at_10": round(rec_all, 4),
            "recall_at_10_strict_first10": round(rec_strict, 4),
            "max_rel_dist_err": float("%.3g" % rel),
            "iters": last["n_iters_run"],
            "stage_ms_per_step": {s: round(v / steps, 3) for s, v in stage.items()},
            "last_step_iter_ms": {"sample": [round(v, 3) for v in last["ms_sample"]],
                                  "join": [round(v, 3) for v in last["ms_join"]],
                                  "merge": [round(v, 3) for v in last["ms_merge"]]},
            "counts": {"leaves": last["n_leaves"], "tree_levels": last["tree_levels"],
                       "leaf_pairs": last["leaf_pairs"], "join_pairs": last["join_pairs"],
                       "join_rows": last["join_rows"], "proposals": last["proposals"], "updates": last["updates"]},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(result))
        sys.stdout.flush()
    builder.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def builder_dp(d):
    return (d + 31) // 32 * 32


if __name__ == "__main__":
    main()

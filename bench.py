"""bench.py -- index-build throughput of the MI355X NN-Descent builder (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete index build (prep -> RP forest -> leaf seeding -> random fill ->
NN-descent to the reference's stop rule -> exact-distance finalize) over a synthetic point set that
is already resident in HBM, with the neighbour graph left resident in HBM.  Workload at N=1:
BASELINE.json configs[1] -- "SIFT-1M (1e6 x 128 float32) euclidean k=15, RP-tree init n_trees=8" --
on the seeded SIFT-like stand-in of SURVEY.md section 8d (no dataset files / network here).

N > 1: one rank per GPU over RCCL (`backend="nccl"`).  The driver launches the ranks with
torch.distributed.run; when WORLD_SIZE is not set (plain `python bench.py --gpus N`) this script re-executes
itself under torch.distributed.run with N ranks and fails loudly if fewer than N devices are visible.
Every N > 1 builds BASELINE configs[3] -- ONE 10 M x 128 set row-sharded over the N GPUs (10 M / N rows per rank) --
"scaling": "strong", so that the curve over N = 2, 4, 8 is over the same work (`one_gpu_same_set` gives its one-GPU time);
--points-per-gpu P selects weak scaling instead (P points per GPU, one global index of N * P points).

Rank 0 prints ONE JSON line (contract in the task statement) carrying, besides the metric:
  roofline             : dominant kernel's algorithmic HBM bytes / its HIP-event duration vs 8 TB/s, plus
                         roofline.mfma: MFMA flops issued (counted by the kernels) vs the f32 matrix peak
  cpu_baseline         : the CPU oracle (restatement of the reference algorithm, oracle/) timed on this
                         box's host cores on the same 1 M points
  value_host_inclusive : the same build through nnd_build (host buffers in, host buffers out: H2D + build + D2H)
  workload_hard        : a second, slow-converging input (latent dimension 48) so tuning is not judged on one workload
                         (+ oracle_recall_at_10: the CPU oracle on the same points and rows, outside the timed region)
  value_class_api      : wall time of the drop-in call itself, pynndescent_amd.NNDescent(x, ...).neighbor_graph (SURVEY 8d's metric)
  workload_k30         : the same points with the reference's default n_neighbors = 30 (+ oracle_recall_at_10, as above)
  workload_c3 / _c5    : BASELINE configs[2] / configs[4] stand-ins (+ oracle_recall_at_10; _c5: + the graph diversification / prune pass)
  roofline.build       : whole-build algorithmic bytes (SURVEY 8d model with the measured counts) / wall time vs 8 TB/s
  recall_at_10         : recall vs exact brute force on a sample of points (reference-test convention)

--data FILE (.fvecs / .npy / .hdf5 with a "train" dataset): build THAT point set instead of the synthetic one (real SIFT-1M /
GloVe / NYTimes when a file is on disk; there is none in this image), N = 1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402  (device memory, streams, torch.distributed: plumbing only)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # same guide: f32-input MFMA (v_mfma_f32_16x16x4_f32) = the f32 vector rate
MFMA_FLOP = 2048.0        # one v_mfma_f32_16x16x4_f32: 16 * 16 * 4 multiply-adds


def sift_like(n, d, seed, device, latent=16, n_clusters=1024, noise=0.3, sample_seed=None):
    """SURVEY.md section 8d C2': 1024-component Gaussian mixture in a 16-dim latent, random linear map to
    d dims, + noise, shifted non-negative, scaled to ~[0, 218]; generated on the device.
    ``seed`` fixes the mixture (centres, projection); ``sample_seed`` the draws (rank-specific when sharded,
    so that all shards come from ONE distribution and neighbours cross shard boundaries)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    centres = torch.randn(n_clusters, latent, generator=g, device=device) * 3.0
    proj = torch.randn(latent, d, generator=g, device=device) / (latent ** 0.5)
    if sample_seed is not None:
        g.manual_seed(sample_seed)
    assign = torch.randint(0, n_clusters, (n,), generator=g, device=device)
    z = centres[assign] + torch.randn(n, latent, generator=g, device=device)
    x = z @ proj + noise * torch.randn(n, d, generator=g, device=device)
    x = x + 12.0  # shift non-negative with a fixed offset (identical on every shard)
    x = x.clamp_min(0.0) * (218.0 / 24.0)
    return x.contiguous().float()


def exact_knn_sample(x, rows, k):
    """Brute-force ground truth (self included) for a sample of rows: f32 Gram pre-selection in chunks of 1M columns
    (torch.topk over rows of several million columns returned wrong answers on this stack), float64 refinement of
    the best 4k candidates."""
    q = x[rows]
    best_v = best_i = None
    for c0 in range(0, x.shape[0], 1_000_000):
        xs = x[c0:c0 + 1_000_000]
        d2 = (q * q).sum(1, keepdim=True) + (xs * xs).sum(1)[None, :] - 2.0 * (q @ xs.T)
        tk = d2.topk(min(4 * k, xs.shape[0]), dim=1, largest=False)
        best_v = tk.values if best_v is None else torch.cat([best_v, tk.values], 1)
        best_i = tk.indices + c0 if best_i is None else torch.cat([best_i, tk.indices + c0], 1)
    cand = torch.gather(best_i, 1, best_v.argsort(dim=1)[:, :4 * k])
    qq = q.double()[:, None, :]
    dd = ((qq - x[cand].double()) ** 2).sum(-1)
    order = dd.argsort(dim=1)[:, :k]
    return torch.gather(cand, 1, order)


def clustered_gpu(n, d, latent, seed, device, nonneg):
    """SURVEY.md section 8d generator of the other BASELINE configurations (C3': latent 24, seed 2; C5': latent 32, seed 4)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    centres = torch.randn(1024, latent, generator=g, device=device) * 3.0
    proj = torch.randn(latent, d, generator=g, device=device) / latent ** 0.5
    assign = torch.randint(0, 1024, (n,), generator=g, device=device)
    x = (centres[assign] + torch.randn(n, latent, generator=g, device=device)) @ proj
    x = x + 0.3 * torch.randn(n, d, generator=g, device=device)
    if nonneg:
        x = (x + 12.0).clamp_min(0) * 9.0
    return x.contiguous()


def exact_knn_cosine(x, rows, k):
    """Exact cosine k-NN (self included) of the sampled rows: f32 pre-selection in chunks of 1M columns, float64 refinement."""
    xn = x / x.norm(dim=1, keepdim=True)
    q = xn[rows]
    best_v = best_i = None
    for c0 in range(0, x.shape[0], 1_000_000):
        dch = 1.0 - q @ xn[c0:c0 + 1_000_000].T
        tk = dch.topk(min(4 * k, dch.shape[1]), dim=1, largest=False)
        best_v = tk.values if best_v is None else torch.cat([best_v, tk.values], 1)
        best_i = tk.indices + c0 if best_i is None else torch.cat([best_i, tk.indices + c0], 1)
    cand = torch.gather(best_i, 1, best_v.argsort(dim=1)[:, :4 * k])
    qd, nb = x[rows].double(), x[cand].double()
    dd = 1.0 - (qd[:, None, :] * nb).sum(-1) / (qd.norm(dim=1)[:, None] * nb.norm(dim=2))
    return torch.gather(cand, 1, dd.argsort(dim=1)[:, :k])


def gather_rows(x, world):
    """All ranks' row blocks (row counts may differ by one), concatenated in rank order, on every rank."""
    import torch.distributed as dist

    host = dist.get_backend() == "gloo"  # (two processes sharing one GPU in the tests: staged through the host)
    t = x.cpu() if host else x
    cnt = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cnts = [int(c.item()) for c in cnts]
    pad = torch.zeros((max(cnts),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:c] for o, c in zip(out, cnts)], dim=0).to(x.device)


def other_config(_capi, name, device, local_rank, oracle=None, oracle_threads=64):
    """BASELINE configs[2] / configs[4] stand-ins (SURVEY 8d C3' / C5'): build time (device resident, best of 2 after a
    warm-up), recall@10 against exact cosine search, and for C5' the graph diversification / prune pass."""
    n, d, latent, seed, k, n_trees = {"c3": (1_200_000, 100, 24, 2, 15, 12), "c5": (290_000, 256, 32, 4, 15, 11)}[name]
    x = clustered_gpu(n, d, latent, seed, device, False)
    lim = np.iinfo(np.int32)
    rs = np.random.RandomState(1)
    rng_state = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64)
    _ = rs.randint(lim.min + 1, lim.max - 1, 3)
    ts = rs.randint(lim.min + 1, lim.max - 1, size=(n_trees, 3)).astype(np.int64)
    b = _capi.Builder(n, d, _capi.NND_METRIC_ALT_COSINE, k, n_trees, max(60, min(256, 5 * k)), 200, min(60, k),
                      max(5, int(round(np.log2(n)))), 0.001, rng_state, ts[0], device=local_rank)
    oi = torch.empty((n, k), dtype=torch.int32, device=device)
    od = torch.empty((n, k), dtype=torch.float32, device=device)
    torch.cuda.synchronize()
    best = None
    for rep_ in range(3):
        b.set_data_device(x.data_ptr(), keepalive=x)
        b.synchronize()
        t1 = time.perf_counter()
        b.build_device(oi.data_ptr(), od.data_ptr())
        b.synchronize()
        dt = time.perf_counter() - t1
        if rep_ > 0 and (best is None or dt < best):
            best = dt
    st = b.stats()
    prune = None
    if name == "c5":
        # the graph diversification / prune pass of configs[4] on the handle that has just built the graph: the k-NN graph stays in
        # HBM, every step between the kernels runs on the device (csrc/searchgraph.hip), ONE copy of indptr / indices comes back
        from pynndescent_amd.search_graph import search_graph_on

        bestp = sg = None
        for rep_ in range(3):
            t1 = time.perf_counter()
            sg, stg = search_graph_on(b, oi.data_ptr(), od.data_ptr(), k, on_device=True, return_stages=True)
            dtp = time.perf_counter() - t1
            if rep_ > 0 and (bestp is None or dtp < bestp[0]):
                bestp = (dtp, stg["ms_device"])
        prune = {"ms": round(bestp[0] * 1e3, 2), "ms_device": round(bestp[1], 2), "edges_in": int((oi >= 0).sum().item()), "edges_out": int(sg.nnz),
                 "max_degree": int(np.diff(sg.indptr).max()),
                 "what": "diversify + reverse diversify + union + degree prune + binarise (pynndescent_.py:1451-1611) on the device, on the handle "
                         "that built the graph: device arrays in, ONE device-to-host copy of the CSR pattern out (incl. the numpy / scipy wrap)"}
    b.close()
    rows_np = np.random.RandomState(0).choice(n, 1000, replace=False)
    rows = torch.from_numpy(rows_np).to(device)
    true10 = exact_knn_cosine(x, rows, 10)
    rec = recall_at(true10, oi[rows], 10)
    leg = None
    if oracle is not None:  # the reference algorithm (CPU oracle) on the same points, recall on the same rows: outside every timed region
        t1 = time.perf_counter()
        oidx, _ = oracle.build_index(x.cpu().numpy(), "cosine", n_neighbors=k, n_trees=n_trees, random_state=1234, n_threads=oracle_threads, kind="fast")
        leg = {"oracle_recall_at_10": round(float(oracle.recall(true10.cpu().numpy(), oidx[rows_np])), 4),
               "oracle_seconds": round(time.perf_counter() - t1, 1), "oracle_threads": oracle_threads}
        leg["recall_gap_to_oracle"] = round(rec - leg["oracle_recall_at_10"], 4)
    out = {"workload": {"c3": "BASELINE configs[2] stand-in (GloVe-like, SURVEY 8d C3'): %dx%d float32 cosine k=%d n_trees=%d",
                        "c5": "BASELINE configs[4] stand-in (NYTimes-like, SURVEY 8d C5'): %dx%d float32 cosine k=%d n_trees=%d + "
                              "graph diversification / prune pass"}[name] % (n, d, k, n_trees),
           "value": round(n / best, 1), "ms_per_step": round(best * 1e3, 3), "iters": st["n_iters_run"], "recall_at_10": round(rec, 4),
           "stage_ms": {"forest": round(st["ms_forest"], 3), "leaf_init": round(st["ms_leaf_init"], 3), "join": round(sum(st["ms_join"]), 3),
                        "sample": round(sum(st["ms_sample"]), 3), "merge": round(sum(st["ms_merge"]), 3), "finalize": round(st["ms_finalize"], 3)}}
    if leg is not None:
        out.update(leg)
    if name == "c5":
        from pynndescent_amd.search_graph import build_search_graph

        xh, gi, gd = x.cpu().numpy(), oi.cpu().numpy(), od.cpu().numpy()
        build_search_graph(xh, gi, gd, "cosine", k)  # warm-up (handle allocation)
        t1 = time.perf_counter()
        sg2 = build_search_graph(xh, gi, gd, "cosine", k)
        prune["ms_host_arrays"] = round((time.perf_counter() - t1) * 1e3, 2)
        prune["what_host_arrays"] = ("the same pass as a stand-alone call on HOST arrays (what NNDescent.prepare() runs): auxiliary handle, "
                                     "H2D of the %d MB point set and of the graph, the pass, D2H of the pattern" % (xh.nbytes // 1000000))
        assert sg2.nnz == prune["edges_out"]
        out["prune_pass"] = prune
    return out


def recall_at(true_idx, approx_idx, k_true=10, cols=None):
    t = true_idx[:, :k_true].cpu().numpy()
    a = approx_idx.cpu().numpy() if cols is None else approx_idx[:, :cols].cpu().numpy()
    hits = sum(np.isin(tr, ar).sum() for tr, ar in zip(t, a))
    return hits / float(t.shape[0] * k_true)


def cpu_baseline(O, xs, k, n_trees, true_rows=None, true_idx=None):
    """The CPU oracle (restatement of the reference algorithm) on this box's host cores, on the SAME points as the
    GPU build.  The number of vertex blocks (the reference's numba thread count, utils.py:259-273) decides both the
    result and how much of the box is busy: it is picked by a probe on the first 200 k points, then the whole set is
    timed once.  (Round 4: the port no longer makes every thread scan all edges / all updates as the reference does --
    same pushes in the same order, pinned bit-exact against the reference -- so it is a STRONGER baseline than the
    reference's own scheme would be on many cores.)"""
    cores = os.cpu_count() or 1
    O.build()
    kind = O.build_native()  # -march=native on THIS box's cores when gcc is here (round-5 review: the shipped library is x86-64-v3)
    probe = xs[: min(200_000, xs.shape[0])]
    best_t, best = None, None
    for t in sorted({min(cores, c) for c in (32, 64, 128, 256, cores)}):
        t1 = time.perf_counter()
        O.build_index(probe, "euclidean", n_neighbors=k, n_trees=n_trees, random_state=1234, n_threads=t, kind=kind)
        dt = time.perf_counter() - t1
        if best is None or dt < best:
            best_t, best = t, dt
    t1 = time.perf_counter()
    oidx, _ = O.build_index(xs, "euclidean", n_neighbors=k, n_trees=n_trees, random_state=1234, n_threads=best_t, kind=kind)
    dt = time.perf_counter() - t1
    rec = None
    if true_rows is not None:
        rec = round(float(O.recall(true_idx, oidx[true_rows])), 4)
    return {"value": round(xs.shape[0] / dt, 1), "unit": "points/s", "cores": best_t, "host_cores": cores, "kind": "port",
            "seconds": round(dt, 2), "recall_at_10": rec, "build_flags": "-O3 -ffast-math -fopenmp -march=%s" % ("native (compiled on this box)" if kind == "native" else "x86-64-v3"),
            "note": "PORT: a C/OpenMP restatement of the reference algorithm (oracle/), NOT the reference's numba code "
                    "(numba is not installable in this image)",
            "sample": "all %d points of the same set, same k/n_trees/defaults; gcc -O3 -ffast-math + OpenMP, %d threads "
                      "(fastest of a probe over 32/64/128/256/all cores on the first %d points); the port buckets edges and "
                      "updates by owner block instead of letting every thread scan all of them (bit-identical results)"
                      % (xs.shape[0], best_t, probe.shape[0])}


def oracle_leg(O, xs, k, n_trees, n_threads, true_rows, true_idx):
    """recall@10 of the CPU oracle (the reference algorithm restated, oracle/) on the SAME points and the SAME sampled rows as
    a GPU workload of this line -- outside every timed region, like cpu_baseline: the two-sided parity check (|GPU - reference
    algorithm| <= 0.005, north star) stated in the bench line itself."""
    t1 = time.perf_counter()
    oidx, _ = O.build_index(xs, "euclidean", n_neighbors=k, n_trees=n_trees, random_state=1234, n_threads=n_threads, kind="fast")
    dt = time.perf_counter() - t1
    return {"oracle_recall_at_10": round(float(O.recall(true_idx, oidx[true_rows])), 4), "oracle_seconds": round(dt, 1), "oracle_threads": n_threads}


def host_inclusive(_capi, x_host, k, n_trees, leaf_size, n_iters, rng_state, tree_state, device, reps=2):
    """n / wall(build -> neighbor_graph arrays on the host): ONE nnd_build call per repetition --
    the entry point INTEGRATION.md binds -- incl. handle creation, H2D of the points and D2H of the graph."""
    import ctypes as C

    lib = _capi.load_library()
    n, d = x_host.shape
    p = _capi.NNDParams()
    p.n, p.dim, p.metric, p.n_neighbors, p.n_trees, p.leaf_size = n, d, 0, k, n_trees, leaf_size
    p.max_depth, p.max_candidates, p.n_iters, p.delta, p.device, p.join_blocks = 200, min(60, k), n_iters, 0.001, device, 0
    for i in range(3):
        p.rng_state[i], p.tree_rng[i] = int(rng_state[i]), int(tree_state[i])
    idx = np.empty((n, k), np.int32)
    dist = np.empty((n, k), np.float32)
    err = C.create_string_buffer(512)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = lib.nnd_build(C.byref(p), x_host.ctypes.data_as(C.c_void_p), None, None, 0, idx.ctypes.data_as(C.c_void_p),
                           dist.ctypes.data_as(C.c_void_p), None, err, 512)
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(err.value.decode())
        best = dt if best is None else min(best, dt)
    return {"value": round(n / best, 1), "unit": "points/s", "ms": round(best * 1e3, 2),
            "what": "nnd_build: hipMalloc of the state + H2D %d MB (pageable numpy) + build + D2H %d MB, best of %d"
                    % (x_host.nbytes >> 20, (idx.nbytes + dist.nbytes) >> 20, reps)}


def class_api(x_host, k, n_trees, device, reps=3):
    """SURVEY.md section 8d's metric, literally: n / wall(NNDescent(x, ...) -> neighbor_graph arrays on the host), warm.
    Measured twice: in THIS process (which has allocated and released many GB by now: the result arrays then come out of
    fragmented 4 KB pages and their first touch costs tens of ms) and in a FRESH process (a user's script: the same call
    after one warm-up call), the latter being the figure reported as `ms`."""
    import pynndescent_amd

    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        idx, dist = pynndescent_amd.NNDescent(x_host, "euclidean", n_neighbors=k, n_trees=n_trees, random_state=1234,
                                              device=device).neighbor_graph
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    n, d = x_host.shape
    fresh = None
    try:
        code = ("import sys, time, torch, numpy as np; sys.path.insert(0, %r); import bench, pynndescent_amd\n"
                "x = bench.sift_like(%d, %d, seed=1, device=torch.device('cuda', %d), sample_seed=100).cpu().numpy()\n"
                "best = None\n"
                "for r in range(%d):\n"
                "    t0 = time.perf_counter(); g = pynndescent_amd.NNDescent(x, 'euclidean', n_neighbors=%d, n_trees=%d, random_state=1234, device=%d).neighbor_graph\n"
                "    dt = time.perf_counter() - t0\n"
                "    if r > 0: best = dt if best is None else min(best, dt)\n"
                "print('CLASS_API_MS', best * 1e3)\n") % (ROOT, n, d, device, reps + 1, k, n_trees, device)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
        for line in out.splitlines():
            if line.startswith("CLASS_API_MS"):
                fresh = float(line.split()[1]) * 1e-3
    except Exception:  # the in-process figure stands
        fresh = None
    use = fresh if fresh is not None else best
    return {"value": round(n / use, 1), "unit": "points/s", "ms": round(use * 1e3, 2), "ms_in_this_process": round(best * 1e3, 2),
            "measured_in": "a fresh process (one warm-up call, best of %d)" % reps if fresh is not None else "this process",
            "what": "pynndescent_amd.NNDescent(x, 'euclidean', n_neighbors=%d, n_trees=%d).neighbor_graph: check_array + handle "
                    "creation + H2D + build (host-driven iteration loop) + D2H + sqrt correction, best of %d" % (k, n_trees, reps)}


def load_points(path):
    """float32 (n, d) from .fvecs (TEXMEX: int32 d, then d floats, per row), .npy, or .hdf5/.h5 with a 'train' dataset."""
    if path.endswith(".fvecs"):
        raw = np.fromfile(path, dtype=np.int32)
        d = int(raw[0])
        return np.ascontiguousarray(raw.reshape(-1, d + 1)[:, 1:].view(np.float32))
    if path.endswith(".npy"):
        return np.ascontiguousarray(np.load(path), dtype=np.float32)
    if path.endswith((".hdf5", ".h5")):
        try:
            import h5py
        except ImportError:
            sys.exit("bench.py: --data %s needs h5py, which is not installed in this image; convert the 'train' dataset to "
                     ".npy or .fvecs" % path)
        with h5py.File(path, "r") as f:
            return np.ascontiguousarray(f["train"][:], dtype=np.float32)
    sys.exit("bench.py: --data expects a .fvecs, .npy or .hdf5 file")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def builder_dp(d):
    return (d + 31) // 32 * 32


class Watchdog:
    """N > 1 only.  A scaling run must end with ONE JSON line whatever happens: when a phase (communicator creation, a
    build, the one-GPU build of the same set) does not come back within its limit, when another rank dies (the launcher
    sends SIGTERM) or when this rank raises, rank 0 prints a line with the contract's fields, `value` 0 and an `error`
    that names the phase -- instead of a hang that the driver has to kill without a line."""

    def __init__(self, rank, world, args):
        import signal
        import threading

        self.rank, self.world, self.args = rank, world, args
        self.name, self.deadline, self.done = "start", None, False
        self.extra = {}
        self._lock = threading.Lock()
        t = threading.Thread(target=self._run, daemon=True)
        t.start()
        try:
            signal.signal(signal.SIGTERM, lambda *_: self.fail("terminated by the launcher (another rank failed or the run was cancelled)"))
        except ValueError:  # not the main thread
            pass

    def phase(self, name, seconds):
        seconds *= float(os.environ.get("PYNND_BENCH_WATCHDOG_SCALE", "1"))  # (tests shrink the limits)
        with self._lock:
            self.name, self.deadline = name, time.monotonic() + seconds

    def finish(self):
        with self._lock:
            self.done, self.deadline = True, None

    def _run(self):
        while True:
            time.sleep(0.5)
            with self._lock:
                late = (not self.done) and self.deadline is not None and time.monotonic() > self.deadline
            if late:
                self.fail("no progress: the phase did not finish within its limit")

    def fail(self, why, code=3):
        with self._lock:
            if self.done:
                return
            self.done = True
        msg = "%s [rank %d, phase: %s]" % (why, self.rank, self.name)
        sys.stderr.write("bench.py: " + msg + "\n")
        sys.stderr.flush()
        if self.rank == 0:
            line = {"metric": "index build: points indexed/sec (recall@10 vs brute force reported alongside)", "value": 0.0, "unit": "points/s",
                    "n_gpus": self.world, "steps": self.args.steps, "warmup": self.args.warmup, "ms_per_step": None, "higher_is_better": True,
                    "scaling": "strong" if self.args.n is None else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": "BASELINE configs[3] stand-in (row-sharded build over %d GPUs): NOT MEASURED, see error" % self.world},
                    "error": msg}
            line.update(self.extra)
            print(json.dumps(line))
            sys.stdout.flush()
        os._exit(code)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-one-gpu", action="store_true", help="N > 1: skip the one-GPU build of the same set (one_gpu_same_set / speedup)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points-per-gpu", dest="n", type=int, default=None)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--k", type=int, default=15)
    ap.add_argument("--n-trees", type=int, default=None)
    ap.add_argument("--join-blocks", type=int, default=0, help="0 (default): the library's schedule, as the drop-in class runs it")
    ap.add_argument("--latent", type=int, default=16, help="latent dimension of the synthetic mixture (48 = the hard workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip value_host_inclusive, value_class_api, workload_hard, workload_k30")
    ap.add_argument("--data", default=None, help="build this point set (.fvecs / .npy / .hdf5 'train') instead of the synthetic one")
    args = ap.parse_args()

    share_gpu = os.environ.get("PYNND_BENCH_SHARE_GPU") == "1"  # debug: all ranks on GPU 0 over gloo
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become N ranks, one per GPU (torch.distributed.run, RCCL)
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not share_gpu:
            sys.exit("bench.py: --gpus %d requested but only %d HIP device(s) are visible" % (args.gpus, ndev))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    wd = Watchdog(rank, world, args) if world > 1 else None
    try:
        run(args, world, rank, local_rank, share_gpu, wd)
    except SystemExit:
        raise
    except BaseException as e:  # N > 1: the line with the error instead of a traceback only
        if wd is None:
            raise
        import traceback

        traceback.print_exc()
        wd.fail("%s: %s" % (type(e).__name__, e), code=1)


def run(args, world, rank, local_rank, share_gpu, wd):
    def phase(name, seconds):
        if wd is not None:
            wd.phase(name, seconds)

    phase("process group", 300)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            if torch.cuda.device_count() <= local_rank:
                sys.exit("bench.py: rank %d has no HIP device %d (visible: %d)" % (rank, local_rank, torch.cuda.device_count()))
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    from pynndescent_amd import _capi, sharded

    d, k = args.dim, args.k
    # N > 1: BASELINE configs[3] -- ONE 10 M x 128 set -- at every N (strong scaling: the curve over N = 2, 4, 8 is over the
    # same work, and `one_gpu_same_set` gives its one-GPU time); --n selects weak scaling with n points per GPU instead
    strong = world > 1 and args.n is None
    if strong:
        n_total = 10_000_000
        lo, hi = sharded.shard_ranges(n_total, world)[rank]
        n = hi - lo
        shard_sizes = [b - a for a, b in sharded.shard_ranges(n_total, world)]
    else:
        n = 1_000_000 if args.n is None else args.n
        n_total = n * world
        shard_sizes = [n] * world
    n_trees = args.n_trees
    if n_trees is None and args.data is None:  # configs[1] names 8 trees; configs[3] names none -> the reference default (pynndescent_.py:1009-1010)
        n_trees = max(3, min(12, int(round(2.0 * np.log10(n_total))))) if strong else 8
    data_name = None
    if args.data is not None:
        if world != 1:
            sys.exit("bench.py: --data is a single-GPU workload")
        xh_file = load_points(args.data)
        n, d = xh_file.shape
        n_total, shard_sizes = n, [n]
        x = torch.from_numpy(xh_file).to(device)
        data_name = os.path.basename(args.data)
        del xh_file
    else:
        x = sift_like(n, d, seed=1, device=device, sample_seed=100 + rank, latent=args.latent)
    torch.cuda.synchronize()
    if n_trees is None and args.data is not None:
        n_trees = max(3, min(12, int(round(2.0 * np.log10(n_total)))))
    n_iters = max(5, int(round(np.log2(n_total))))  # pynndescent_.py:1011-1012
    leaf_size = max(60, min(256, 5 * k))              # rp_trees.py:2845-2846
    lim = np.iinfo(np.int32)
    rs = np.random.RandomState(1234)
    rng_state = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64)
    _ = rs.randint(lim.min + 1, lim.max - 1, 3)
    tree_states = rs.randint(lim.min + 1, lim.max - 1, size=(n_trees, 3)).astype(np.int64)

    if world == 1:
        out_idx = torch.empty((n, k), dtype=torch.int32, device=device)
        out_dist = torch.empty((n, k), dtype=torch.float32, device=device)
        builder = _capi.Builder(n, d, _capi.NND_METRIC_SQEUCLIDEAN, k, n_trees, leaf_size, 200, min(60, k), n_iters,
                                0.001, rng_state, tree_states[0], device=local_rank, join_blocks=args.join_blocks)
        sb = None

        def step():
            builder.set_data_device(x.data_ptr(), keepalive=x)
            builder.build_device(out_idx.data_ptr(), out_dist.data_ptr())
            return builder.stats(), None
    else:
        # the rank's communicator: RCCL (nccl backend; only the 128-byte unique id travels through torch) -- the exchanges
        # themselves are issued by libpynnd_amd.so (ncclGroupStart / ncclSend / ncclRecv) on the build's HIP stream
        phase("communicator (RCCL: two channels, first exchange on each)", 300)
        rccl_error = None
        try:
            comm = sharded.make_comm(local_rank, allow_host_fallback=False, timeout_s=300)  # a scaling run never becomes a host-staged one silently
        except _capi.NNDError as e:  # (raised on EVERY rank: the ranks agree on the outcome, sharded.make_comm)
            # ... but a LOUD host-staged number beats no line at all: config.comm.transport says "host", `degraded` says why
            rccl_error = str(e)
            if rank == 0:
                sys.stderr.write("bench.py: %s -- falling back to the HOST transport (pinned staging + gloo): NOT a scaling number\n" % rccl_error)
            comm = sharded.make_comm(local_rank, allow_host_fallback=True, timeout_s=300)
        if wd is not None:
            wd.extra["config"] = {"workload": "BASELINE configs[3] stand-in (row-sharded build over %d GPUs): NOT MEASURED, see error" % world,
                                  "comm": comm.info()}
        phase("shard allocation", 300)
        sb = sharded.ShardedBuilder(comm, shard_sizes, d, "euclidean", k, n_trees, seed=1234, device_index=local_rank)
        builder = None
        out_idx, out_dist = sb.out_idx, sb.out_dist

        def step():
            _, _, info = sb.build(x)
            return info["stats"], info

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        if builder is not None:
            builder.synchronize()

    for i in range(args.warmup):
        phase("warm-up build %d" % i, 600)
        if os.environ.get("PYNND_BENCH_TEST_STALL_RANK") == str(rank):  # test hook: this rank never enters its build
            time.sleep(3600)
        step()
    phase("barrier before the timed builds", 300)
    barrier()
    phase("timed builds", 120 + 120 * args.steps)
    t0 = time.perf_counter()
    stage = {"forest": 0.0, "leaf_init": 0.0, "join": 0.0, "sample": 0.0, "merge": 0.0, "finalize": 0.0, "prep": 0.0,
             "random_init": 0.0}
    join_bytes = join_ms = leaf_bytes = 0.0
    join_mfma = leaf_mfma = join_pairs = leaf_pairs = 0.0
    n_join_launches = 0
    last = info = None
    for _ in range(args.steps):
        st, info = step()  # per-stage HIP-event timings are taken on the library's own stream
        last = st
        for key, name in (("forest", "ms_forest"), ("leaf_init", "ms_leaf_init"), ("finalize", "ms_finalize"),
                          ("prep", "ms_prep"), ("random_init", "ms_random_init")):
            stage[key] += st[name]
        stage["join"] += sum(st["ms_join"])
        stage["sample"] += sum(st["ms_sample"])
        stage["merge"] += sum(st["ms_merge"])
        join_ms += sum(st["ms_join"])
        join_bytes += sum(st["join_rows"]) * 4.0 * builder_dp(d)
        leaf_bytes += st["leaf_rows"] * 4.0 * builder_dp(d)
        join_mfma += sum(st["join_mfma"])
        leaf_mfma += st["leaf_mfma"]
        join_pairs += sum(st["join_pairs"])
        leaf_pairs += st["leaf_pairs"]
        n_join_launches += sum(st.get("join_substeps") or [args.join_blocks or 1] * st["n_iters_run"])
    barrier()
    elapsed = time.perf_counter() - t0
    phase("recall / one-GPU build of the same set / extras", 1500)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1000.0 / args.steps
    value = n_total * args.steps / elapsed

    # recall vs exact brute force on a sample of this rank's rows (outside the timed region)
    x_all = gather_rows(x, world) if world > 1 else x
    one_gpu = None
    if world > 1 and rank == 0 and not args.no_one_gpu:
        # the SAME set on one GPU (rank 0, outside the timed region): what the N-GPU time is a speed-up OVER
        b1 = _capi.Builder(n_total, d, _capi.NND_METRIC_SQEUCLIDEAN, k, n_trees, leaf_size, 200, min(60, k), n_iters, 0.001, rng_state,
                           tree_states[0], device=local_rank)
        o1i = torch.empty((n_total, k), dtype=torch.int32, device=device)
        o1d = torch.empty((n_total, k), dtype=torch.float32, device=device)
        torch.cuda.synchronize()
        b1.set_data_device(x_all.data_ptr(), keepalive=x_all)
        b1.build_device(o1i.data_ptr(), o1d.data_ptr())  # warm-up
        b1.synchronize()
        t1 = time.perf_counter()
        b1.set_data_device(x_all.data_ptr(), keepalive=x_all)
        b1.build_device(o1i.data_ptr(), o1d.data_ptr())
        b1.synchronize()
        dt1 = time.perf_counter() - t1
        rs1 = torch.from_numpy(np.random.RandomState(0).choice(n, size=min(2000, n), replace=False)).to(device)
        one_gpu = {"ms": round(dt1 * 1e3, 3), "value": round(n_total / dt1, 1), "iters": b1.stats()["n_iters_run"],
                   "recall_at_10": round(recall_at(exact_knn_sample(x_all, rs1, 10), o1i[rs1], 10), 4),
                   "what": "the same %d points built by ONE GPU (rank 0's), same parameters, timed once after a warm-up build" % n_total}
        b1.close()
        del o1i, o1d
    if rank == 0:
        rsmp = np.random.RandomState(0)
        rows_np = rsmp.choice(n, size=min(2000, n), replace=False)
        rows = torch.from_numpy(rows_np).to(device)  # rank 0 owns rows [0, n)
        true_idx = exact_knn_sample(x_all, rows, 10)
        rec_all = recall_at(true_idx, out_idx[rows], 10)
        rec_strict = recall_at(true_idx, out_idx[rows], 10, cols=10)
        nb = x_all[out_idx[rows].long()].double()
        truth = ((x_all[rows].double()[:, None, :] - nb) ** 2).sum(-1)
        rel = ((out_dist[rows].double() - truth).abs() / truth.clamp_min(1e-30))[truth > 0].max().item()

        # roofline of the dominant kernel (by measured time).  Algorithmic bytes:
        #   k_local_join : C_i gathered candidate rows * dp*4 bytes per launch
        #   k_leaf_join  : sum of leaf sizes * dp*4 bytes per launch
        #   rp forest    : n * dp*4 * levels (each row once per level, all trees fused) + 8 B/position/level
        steps = float(args.steps)
        local_trees = max(1, n_trees // world)
        dominant = max(("join", "leaf_init"), key=lambda sname: stage[sname])
        join_gbs = join_bytes / (join_ms * 1e-3) / 1e9 if join_ms > 0 else 0.0
        leaf_gbs = leaf_bytes / (stage["leaf_init"] * 1e-3) / 1e9 if stage["leaf_init"] > 0 else 0.0
        if dominant == "join":
            achieved, kernel = join_gbs, "k_local_join"
        else:
            achieved, kernel = leaf_gbs, "k_leaf_join"
        n_here = n_total
        row = 4.0 * builder_dp(d)
        # Levels = ceil(log2(n / mean leaf fill)) -- NOT the measured level count (the routing forest never
        # reads a row once per level of its deepest branch)
        leaf_fill = local_trees * n_here / max(last["n_leaves"], 1)
        levels = int(np.ceil(np.log2(max(n_here / max(leaf_fill, 1.0), 2.0))))
        tree_bytes = steps * (n_here * row * levels + n_here * 8.0 * levels * local_trees)
        forest_gbs = tree_bytes / (stage["forest"] * 1e-3) / 1e9 if stage["forest"] > 0 else 0.0
        # HBM traffic per launch from the committed PMC passes (tools/pmc_traffic.py; separate --pmc runs, FETCH_SIZE
        # corrected by the factor calibrated for this access pattern); null if no profile of this kernel is committed
        traffic = traffic_from = None
        for tname in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if traffic is None and os.path.exists(tpath):
                tj = json.load(open(tpath))
                key = {"join": "k_local_join", "leaf_init": "k_leaf_join"}[dominant]
                for name, rec in tj.items():
                    if isinstance(rec, dict) and key in name:
                        traffic = rec["traffic_bytes_per_launch"]
                        traffic_from = "profiles/%s (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel; NOT measured in this run)" % tname

        # MFMA side (north_star: "MFMA utilisation reported against gfx950 peak"): instructions are counted by the
        # kernels themselves (v_mfma_f32_16x16x4_f32, 2048 flop each); algorithmic = 2*dp flop per evaluated pair
        def mfma_rec(instr, ms, pairs):
            issued = instr * MFMA_FLOP / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            alg = pairs * 2.0 * builder_dp(d) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"issued_tflops": round(issued, 2), "frac_of_f32_mfma_peak": round(issued / MFMA_F32_PEAK_TF, 4),
                    "algorithmic_tflops": round(alg, 2), "tile_efficiency": round(alg / issued, 3) if issued > 0 else None,
                    "mfma_instructions_per_build": round(instr / steps)}

        mfma = {"peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "instruction": "v_mfma_f32_16x16x4_f32",
                "k_local_join": mfma_rec(join_mfma, join_ms, join_pairs),
                "k_leaf_join": mfma_rec(leaf_mfma, stage["leaf_init"], leaf_pairs)}
        roofline = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_from": traffic_from,
                    "k_local_join": {"achieved": round(join_gbs, 2), "frac": round(join_gbs / HBM_PEAK_GBS, 5),
                                     "avg_launch_ms": round(join_ms / max(n_join_launches, 1), 4),
                                     "bytes_per_launch": round(join_bytes / max(n_join_launches, 1))},
                    "k_leaf_join": {"achieved": round(leaf_gbs, 2), "frac": round(leaf_gbs / HBM_PEAK_GBS, 5),
                                    "avg_launch_ms": round(stage["leaf_init"] / steps / local_trees, 4),
                                    "bytes_per_launch": round(leaf_bytes / steps / local_trees)},
                    "rp_forest_stage": {"achieved": round(forest_gbs, 2), "frac": round(forest_gbs / HBM_PEAK_GBS, 5),
                                        "ms": round(stage["forest"] / steps, 3), "levels_in_model": levels},
                    "mfma": mfma}
        # whole build (single GPU): SURVEY 8d's B = B_tree + B_leaf + sum_i B_iter + B_final with the measured counts,
        # over the wall time of the timed region -- so that the headline fraction is not only the best kernel's
        if world == 1:
            c_rows = float(sum(last["join_rows"]))
            b_tree = n_here * row * levels + n_here * 8.0 * levels * n_trees
            b_leaf = n_trees * n_here * row + n_trees * n_here * 4.0
            b_iter = c_rows * row + last["n_iters_run"] * (2.0 * n_here * k * 9.0 + n_here * k * 5.0) + 2.0 * c_rows * 4.0
            b_final = n_here * k * row + n_here * k * 8.0
            b_all = b_tree + b_leaf + b_iter + b_final
            gbs = b_all / (ms_per_step * 1e-3) / 1e9
            roofline["build"] = {"algorithmic_bytes": round(b_all), "achieved": round(gbs, 2), "frac": round(gbs / HBM_PEAK_GBS, 5),
                                 "parts_gb": {"tree": round(b_tree / 1e9, 2), "leaf": round(b_leaf / 1e9, 2),
                                              "iters": round(b_iter / 1e9, 2), "final": round(b_final / 1e9, 2)}}

        cpu = host_incl = hard = cls_api = k30 = c3 = c5 = hard_leg = None
        if world == 1 and not args.no_extras:
            x_host = x.cpu().numpy()
            host_incl = host_inclusive(_capi, x_host, k, n_trees, leaf_size, n_iters, rng_state, tree_states[0], local_rank)
            cls_api = class_api(x_host, k, n_trees, local_rank)
            del x_host
            if args.latent == 16 and args.data is None:  # the reference's default n_neighbors on the same points
                k3 = 30
                b30 = _capi.Builder(n, d, _capi.NND_METRIC_SQEUCLIDEAN, k3, n_trees, max(60, min(256, 5 * k3)), 200, min(60, k3), n_iters,
                                    0.001, rng_state, tree_states[0], device=local_rank)
                o_i = torch.empty((n, k3), dtype=torch.int32, device=device)
                o_d = torch.empty((n, k3), dtype=torch.float32, device=device)
                b30.set_data_device(x.data_ptr(), keepalive=x)
                b30.build_device(o_i.data_ptr(), o_d.data_ptr())  # warm-up
                b30.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    b30.set_data_device(x.data_ptr(), keepalive=x)
                    b30.build_device(o_i.data_ptr(), o_d.data_ptr())
                b30.synchronize()
                dt = (time.perf_counter() - t1) / 2
                s30 = b30.stats()
                k30 = {"workload": "the same %dx%d points, n_neighbors=30 (the reference's default), n_trees=%d" % (n, d, n_trees),
                       "value": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 3), "iters": s30["n_iters_run"],
                       "recall_at_10": round(recall_at(true_idx, o_i[rows], 10), 4),
                       "stage_ms": {"forest": round(s30["ms_forest"], 3), "leaf_init": round(s30["ms_leaf_init"], 3),
                                    "join": round(sum(s30["ms_join"]), 3), "sample": round(sum(s30["ms_sample"]), 3),
                                    "merge": round(sum(s30["ms_merge"]), 3), "finalize": round(s30["ms_finalize"], 3)}}
                # the same roofline arithmetic as the headline's: algorithmic bytes (candidate rows gathered / leaf rows
                # loaded, x 4 dp) over the stage's HIP-event time, against 8 TB/s
                j_gbs = sum(s30["join_rows"]) * row / (sum(s30["ms_join"]) * 1e-3) / 1e9 if sum(s30["ms_join"]) > 0 else 0.0
                l_gbs = s30["leaf_rows"] * row / (s30["ms_leaf_init"] * 1e-3) / 1e9 if s30["ms_leaf_init"] > 0 else 0.0
                k30["roofline"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "k_local_join_w": {"achieved": round(j_gbs, 2), "frac": round(j_gbs / HBM_PEAK_GBS, 5),
                                                      "bytes_per_build": round(sum(s30["join_rows"]) * row),
                                                      "pairs_per_row": round(sum(s30["join_pairs"]) / max(sum(s30["join_rows"]), 1), 2)},
                                   "k_leaf_join_rb": {"achieved": round(l_gbs, 2), "frac": round(l_gbs / HBM_PEAK_GBS, 5),
                                                      "bytes_per_build": round(s30["leaf_rows"] * row)}}
                b30.close()
                del o_i, o_d
            if args.latent == 16 and args.data is None:  # second perf line: same generator, latent dimension 48 (converges slowly)
                xh = sift_like(n, d, seed=1, device=device, sample_seed=100, latent=48)
                torch.cuda.synchronize()
                builder.set_data_device(xh.data_ptr(), keepalive=xh)
                builder.build_device(out_idx.data_ptr(), out_dist.data_ptr())  # warm-up
                builder.synchronize()
                t1 = time.perf_counter()
                for _ in range(2):
                    builder.set_data_device(xh.data_ptr(), keepalive=xh)
                    builder.build_device(out_idx.data_ptr(), out_dist.data_ptr())
                builder.synchronize()
                dt = (time.perf_counter() - t1) / 2
                sth = builder.stats()
                rows_h = rows[:1000]
                th = exact_knn_sample(xh, rows_h, 10)
                hard_leg = (xh.cpu().numpy(), rows_np[:1000], th.cpu().numpy()) if not args.no_cpu_baseline else None
                hard = {"workload": "same generator, latent dimension 48: %dx%d euclidean k=%d n_trees=%d" % (n, d, k, n_trees),
                        "value": round(n / dt, 1), "ms_per_step": round(dt * 1e3, 3), "iters": sth["n_iters_run"],
                        "recall_at_10": round(recall_at(th, out_idx[rows_h], 10), 4),
                        "stage_ms": {"forest": round(sth["ms_forest"], 3), "leaf_init": round(sth["ms_leaf_init"], 3),
                                     "join": round(sum(sth["ms_join"]), 3), "sample": round(sum(sth["ms_sample"]), 3),
                                     "merge": round(sum(sth["ms_merge"]), 3), "finalize": round(sth["ms_finalize"], 3)}}
                del xh
        if world == 1 and not args.no_extras and args.latent == 16 and args.data is None:
            builder.close()  # (its HBM is not needed any more; the other configurations get the GPU to themselves)
            _capi.load_library().nnd_release_pending()
            orc = None
            if not args.no_cpu_baseline:
                from oracle import oracle as orc  # test infrastructure: the oracle legs only (outside every timed region)
            othr = min(os.cpu_count() or 1, 256)
            c3 = other_config(_capi, "c3", device, local_rank, orc, othr)
            c5 = other_config(_capi, "c5", device, local_rank, orc, othr)
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O  # test infrastructure: the cpu_baseline leg only

            x_np = x.cpu().numpy()
            cpu = cpu_baseline(O, x_np, k, n_trees, rows_np, true_idx.cpu().numpy())
            # an oracle leg under every workload the line reports (round-5 review): the reference algorithm's recall on the same rows
            if k30 is not None:
                k30.update(oracle_leg(O, x_np, 30, n_trees, cpu["cores"], rows_np, true_idx.cpu().numpy()))
                k30["recall_gap_to_oracle"] = round(k30["recall_at_10"] - k30["oracle_recall_at_10"], 4)
            if hard is not None and hard_leg is not None:
                hard.update(oracle_leg(O, hard_leg[0], k, n_trees, cpu["cores"], hard_leg[1], hard_leg[2]))
                hard["recall_gap_to_oracle"] = round(hard["recall_at_10"] - hard["oracle_recall_at_10"], 4)
            del x_np, hard_leg

        if data_name is not None:
            workload = "%s: %dx%d float32 euclidean k=%d n_trees=%d (file given with --data)" % (data_name, n_total, d, k, n_trees)
        elif strong:
            workload = ("BASELINE configs[3] stand-in: ONE SIFT-like %dx%d float32 euclidean k=%d set (n_trees=%d) row-sharded "
                        "over %d GPUs, %d rows per rank" % (n_total, d, k, n_trees, world, n))
        else:
            workload = ("SIFT-like %dx%d float32 euclidean k=%d n_trees=%d (BASELINE configs[1] stand-in, SURVEY 8d C2'%s); "
                        "%d points per GPU" % (n_total, d, k, n_trees, "" if args.latent == 16 else ", latent %d" % args.latent, n))
        result = {
            "metric": "index build: points indexed/sec (recall@10 vs brute force reported alongside)",
            "value": round(value, 1),
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if data_name is None else "file:" + data_name,
            "config": {"workload": workload,
                       "parallelism": "1 GPU" if world == 1 else
                       "rows sharded over %d GPUs (one global index of %d points): point set all-gathered once (second channel, "
                       "behind the forest's first steps), forest %s, per iteration threshold all-gather + reverse-offer "
                       "all-to-all-v + proposal all-to-all-v, the update counts riding on the record-count exchange; "
                       "ncclSend/ncclRecv groups issued by libpynnd_amd.so on the build's stream (%s transport)"
                       % (world, n_total, "sharded by cell" if (info or {}).get("forest_by_cell") else "split by tree",
                          {"rccl": "RCCL", "host": "HOST-staged over gloo (debug: two processes sharing one GPU)"}.get(comm.transport, comm.transport)),
                       "comm": None if world == 1 else comm.info(),
                       "join_blocks": args.join_blocks, "join_substeps_last_step": last.get("join_substeps")},
            "recall_at_10": round(rec_all, 4),
            "recall_at_10_strict_first10": round(rec_strict, 4),
            "max_rel_dist_err": float("%.3g" % rel),
            "iters": last["n_iters_run"],
            "stage_ms_per_step": {sname: round(v / steps, 3) for sname, v in stage.items()},
            "last_step_iter_ms": {"sample": [round(v, 3) for v in last["ms_sample"]],
                                  "join": [round(v, 3) for v in last["ms_join"]],
                                  "merge": [round(v, 3) for v in last["ms_merge"]]},
            "counts": {"leaves": last["n_leaves"], "tree_levels": last["tree_levels"], "leaf_pairs": last["leaf_pairs"],
                       "join_pairs": last["join_pairs"], "join_rows": last["join_rows"], "proposals": last["proposals"],
                       "updates": last["updates"]},
            "exchanged_records_rank0": None if info is None else info["exchanged_records"],
            "shard_rank0": None if info is None else {kk: info[kk] for kk in ("c", "offer_records", "proposal_records", "deferred",
                                                                                "dropped_offers", "bytes_sent", "ms_total",
                                                                                "ms_allgather", "ms_klist_exchange", "local_trees",
                                                                                "forest_by_cell", "forest_positions")},
            "one_gpu_same_set": one_gpu,
            "speedup": None if one_gpu is None else round(one_gpu["ms"] / ms_per_step, 3),
            "roofline": roofline,
            "value_host_inclusive": host_incl,
            "value_class_api": cls_api,
            "workload_hard": hard,
            "workload_k30": k30,
            "workload_c3": c3,
            "workload_c5": c5,
            "cpu_baseline": cpu,
        }
        if world > 1 and rccl_error is not None:
            result["degraded"] = "RCCL unavailable (%s): exchanged through the HOST transport -- functional check, not a scaling number" % rccl_error
        if wd is not None:
            wd.finish()
        print(json.dumps(result))
        sys.stdout.flush()
    if wd is not None:
        wd.finish()
    if sb is not None:
        sb.close()
        comm.close()
    else:
        builder.close()  # (idempotent: the extras may have released it already)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

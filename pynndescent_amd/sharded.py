"""Row-sharded multi-GPU index build: one process (or thread) per GPU, exchanges over RCCL.

The reference is single-process; what it offers as a sharding rule is owner-computes over contiguous
vertex ranges (``apply_graph_update_array`` utils.py:709-731, ``new_build_candidates`` utils.py:259-306,
``init_rp_tree`` pynndescent_.py:154-185).  Here the same rule crosses GPUs:

* rank r owns rows ``[lo_r, hi_r)`` of the k-lists; the point set is replicated once (all-gather), so
  candidate vectors never travel again;
* the RP forest is split by TREE: rank r builds ``n_trees/G`` trees over all points, seeds the k-lists
  of every point from its own leaves, and the partial lists are shipped to the owners
  (all-to-all-v of k-list row blocks) and merged there;
* per NN-descent iteration: (1) all-gather of the k-list rows, so that every rank can offer reverse
  candidates to the vertices it owns by scanning all edges -- exactly the reference's per-thread edge scan
  -- and can test proposals against remote thresholds / neighbour ids; (2) local sampling + join of the
  owned vertices; (3) all-to-all-v of the proposals whose target is owned elsewhere, as (key, target)
  records; (4) owner-side merge; (5) all-reduce of the update count for the stop rule (pynndescent_.py:317).

The device-side halves are C-ABI entry points of ``include/pynnd_amd.h``; this module is host
plumbing (torch tensors as device buffers, ``torch.distributed`` -- backend "nccl" is RCCL on ROCm).
``ThreadComm`` runs the same code with G ranks as threads of one process on one GPU (tests).
"""
import threading

import numpy as np
import torch

from . import _capi


# ------------------------------------------------------------------------------------------------
# partitioning helpers (pure python / torch; covered by the gloo CPU tests)

def shard_ranges(n_total, world):
    """Contiguous, near-equal row ranges: rank r owns [n*r//G, n*(r+1)//G)."""
    return [(n_total * r // world, n_total * (r + 1) // world) for r in range(world)]


def tree_ranges(n_trees, world):
    """Trees are dealt in contiguous runs as well; ranks beyond n_trees get none."""
    return [(n_trees * r // world, n_trees * (r + 1) // world) for r in range(world)]


def segment_bounds(offsets_ext, ranges):
    """Record ranges per destination rank. ``offsets_ext`` is the exclusive scan of the per-vertex record
    counts with the grand total appended (length n+1); records are ordered by target vertex, and ranks own
    contiguous vertex ranges, so rank s receives records [offsets_ext[lo_s], offsets_ext[hi_s])."""
    return [(int(offsets_ext[lo]), int(offsets_ext[hi])) for lo, hi in ranges]


# ------------------------------------------------------------------------------------------------
# communicators

class TorchDistComm:
    """torch.distributed transport (backend nccl == RCCL over xGMI on ROCm; gloo on CPU for tests)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_gather_v(self, t):
        """Gather 1-D/2-D tensors whose first dimension may differ per rank."""
        if self._host_staged() and t.is_cuda:
            return [g.to(t.device) for g in self.all_gather_v(t.cpu())]
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        self.dist.all_gather(sizes, n, group=self.group)
        sizes = [int(s.item()) for s in sizes]
        mx = max(sizes)
        pad = t
        if t.shape[0] < mx:
            pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[: t.shape[0]] = t
        out = [torch.empty_like(pad) for _ in range(self.world)]
        self.dist.all_gather(out, pad.contiguous(), group=self.group)
        return [o[:s] for o, s in zip(out, sizes)]

    def all_to_all_v(self, send):
        """send[s] goes to rank s (first-dimension sizes arbitrary); returns the list received."""
        if self._host_staged() and send[0].is_cuda:
            dev0 = send[0].device
            return [g.to(dev0) for g in self.all_to_all_v([t.cpu() for t in send])]
        dev = send[0].device
        counts = torch.tensor([t.shape[0] for t in send], dtype=torch.int64, device=dev)
        rcounts = torch.empty_like(counts)
        self.dist.all_to_all_single(rcounts, counts, group=self.group)
        rc = [int(c) for c in rcounts.tolist()]
        tail = tuple(send[0].shape[1:])
        recv = [torch.empty((c,) + tail, dtype=send[0].dtype, device=dev) for c in rc]
        if self.dist.get_backend(self.group) == "gloo":  # gloo has no all_to_all for lists on every build: pairwise
            ops = []
            for peer in range(self.world):
                if peer == self.rank:
                    recv[peer].copy_(send[peer])
                    continue
                if send[peer].numel():
                    ops.append(self.dist.isend(send[peer].contiguous(), peer, group=self.group))
                if recv[peer].numel():
                    ops.append(self.dist.irecv(recv[peer], peer, group=self.group))
            for op in ops:
                op.wait()
        else:  # nccl (= RCCL): one all_to_all_single with split sizes, the canonical all-to-all-v
            sc = [int(c) for c in counts.tolist()]
            inp = torch.cat([t.reshape((t.shape[0],) + tail) for t in send], dim=0).contiguous()
            out = torch.empty((sum(rc),) + tail, dtype=send[0].dtype, device=dev)
            self.dist.all_to_all_single(out, inp, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
            recv = list(torch.split(out, rc, dim=0))
        return recv

    def all_reduce_sum(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return int(t.item())

    def _dev(self):
        return torch.device("cuda", torch.cuda.current_device()) if self.dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def _host_staged(self):
        """gloo moves host memory: device tensors are staged through the CPU (tests / debugging only)."""
        return self.dist.get_backend(self.group) == "gloo"

    def barrier(self):
        self.dist.barrier(group=self.group)


class ThreadComm:
    """G ranks as threads of one process (one GPU): same call pattern, exchange through shared lists."""

    class _Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.s = shared
        self.rank = rank
        self.world = shared.world

    @classmethod
    def make(cls, world):
        sh = cls._Shared(world)
        return [cls(sh, r) for r in range(world)]

    def _exchange(self, obj):
        self.s.slots[self.rank] = obj
        self.s.barrier.wait()
        got = list(self.s.slots)
        self.s.barrier.wait()
        return got

    def all_gather_v(self, t):
        return [g.clone() for g in self._exchange(t)]

    def all_to_all_v(self, send):
        allsend = self._exchange(send)
        return [allsend[src][self.rank].clone() for src in range(self.world)]

    def all_reduce_sum(self, value):
        return int(sum(self._exchange(int(value))))

    def barrier(self):
        self.s.barrier.wait()


# ------------------------------------------------------------------------------------------------

def _sync():
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()


class ShardedBuilder:
    """Persistent state of one rank (HBM allocations survive across builds; bench.py times ``build``)."""

    def __init__(self, comm, shard_sizes, dim, metric="euclidean", n_neighbors=15, n_trees=8, leaf_size=None,
                 max_candidates=None, n_iters=None, delta=0.001, seed=0, max_rptree_depth=200, device_index=None):
        self.comm = comm
        rank, world = comm.rank, comm.world
        sizes = [int(v) for v in shard_sizes]
        assert len(sizes) == world
        self.n_total = sum(sizes)
        bounds = np.concatenate([[0], np.cumsum(sizes)])
        self.ranges = [(int(bounds[r]), int(bounds[r + 1])) for r in range(world)]
        self.lo, self.hi = self.ranges[rank]
        self.k = int(n_neighbors)
        self.d = int(dim)
        self.delta = float(delta)
        self.n_trees = int(n_trees)
        # the reference's derived defaults, on the GLOBAL n (pynndescent_.py:1009-1012, 1135-1138; rp_trees.py:2845)
        self.n_iters = max(5, int(round(np.log2(self.n_total)))) if n_iters is None else int(n_iters)
        leaf_size = max(60, min(256, 5 * self.k)) if leaf_size is None else int(leaf_size)
        mc = min(60, self.k) if max_candidates is None else int(max_candidates)
        rs = np.random.RandomState(seed)  # identical draws on every rank
        lim = np.iinfo(np.int32)
        rng_state = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64)
        _search = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64)
        tree_states = rs.randint(lim.min + 1, lim.max - 1, size=(max(self.n_trees, 1), 3)).astype(np.int64)
        t0, t1 = tree_ranges(self.n_trees, world)[rank]
        self.local_trees = t1 - t0
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.dev = torch.device("cuda", device_index)
        metric_code = {"euclidean": _capi.NND_METRIC_SQEUCLIDEAN, "l2": _capi.NND_METRIC_SQEUCLIDEAN,
                       "cosine": _capi.NND_METRIC_ALT_COSINE}[metric]
        self.b = _capi.Builder(self.n_total, self.d, metric_code, self.k, self.local_trees, leaf_size, max_rptree_depth, mc,
                               self.n_iters, delta, rng_state, tree_states[min(t0, max(self.n_trees, 1) - 1)],
                               device=device_index)
        self.b.set_owned_range(self.lo, self.hi)
        self.ks = self.b.row_stride()
        dev, ks = self.dev, self.ks
        self.own_e = torch.empty(((self.hi - self.lo) * ks,), dtype=torch.int32, device=dev)
        self.own_th = torch.empty((self.hi - self.lo,), dtype=torch.float32, device=dev)
        self.cnt = torch.zeros((self.n_total,), dtype=torch.int32, device=dev)
        self.offsets = torch.zeros((self.n_total + 1,), dtype=torch.int64, device=dev)
        self.edge_index = torch.tensor([v for ab in self.ranges for v in ab], device=dev)
        self.out_idx = torch.empty((self.hi - self.lo, self.k), dtype=torch.int32, device=dev)
        self.out_dist = torch.empty((self.hi - self.lo, self.k), dtype=torch.float32, device=dev)

    def close(self):
        self.b.close()

    def build(self, x_local, verbose=False):
        """One complete sharded build.  Returns (idx (n_local,k) GLOBAL ids, alt-space dist, info)."""
        comm, b, dev, ks, k = self.comm, self.b, self.dev, self.ks, self.k
        rank, world = comm.rank, comm.world
        lo, hi, ranges, n_total = self.lo, self.hi, self.ranges, self.n_total
        info = {"n_total": n_total, "range": (lo, hi), "local_trees": self.local_trees, "iters": 0, "c": [],
                "exchanged_records": []}
        # ---- replicate the point set once (all-gather over xGMI): candidate vectors never travel again ----
        if world > 1:
            x_full = torch.cat(comm.all_gather_v(x_local.contiguous()), dim=0).contiguous()
        else:
            x_full = x_local.contiguous()
        assert x_full.shape == (n_total, self.d)
        _sync()
        b.set_data_device(x_full.data_ptr(), keepalive=x_full)  # prep kernel + k-list reset

        # ---- forest split by tree: every rank seeds ALL rows from its own trees, owners merge the partial lists ----
        if self.local_trees > 0:
            b.make_forest()
            b.init_from_leaves()
        if self.n_trees > 0 and world > 1:
            send_e, send_d = [], []
            for (a, z) in ranges:
                e = torch.empty(((z - a) * ks,), dtype=torch.int32, device=dev)
                dd = torch.empty(((z - a) * ks,), dtype=torch.float32, device=dev)
                b.export_graph_rows(a, z, e.data_ptr(), dd.data_ptr())
                send_e.append(e)
                send_d.append(dd)
            recv_e = comm.all_to_all_v(send_e)
            recv_d = comm.all_to_all_v(send_d)
            _sync()
            for src in range(world):
                if src != rank and recv_e[src].numel():
                    b.merge_graph_rows(lo, hi, recv_e[src].data_ptr(), recv_d[src].data_ptr())
            del send_e, send_d, recv_e, recv_d
        b.init_random()  # owned rows that are still not full (pynndescent_.py:188-203)

        for it in range(self.n_iters):
            # (1) k-list all-gather: neighbour words (ids + new flags: dedup and reverse edges) and the per-row
            #     worst distances (thresholds) of remote rows -- 4*ks + 4 bytes per row, not the distance rows
            if world > 1:
                b.export_graph_rows(lo, hi, self.own_e.data_ptr(), None)
                b.export_thresholds(lo, hi, self.own_th.data_ptr())
                all_e = comm.all_gather_v(self.own_e)
                all_t = comm.all_gather_v(self.own_th)
                _sync()
                for src, (a, z) in enumerate(ranges):
                    if src != rank and z > a:
                        b.import_graph_rows(a, z, all_e[src].data_ptr(), None)
                        b.import_thresholds(a, z, all_t[src].data_ptr())
                del all_e, all_t
            # (2) local sampling and join of the owned vertices
            b.descent_sample()
            b.descent_join()
            # (3) proposals for vertices owned elsewhere -> (key, target) records -> owners
            n_sent = 0
            if world > 1:
                b.proposal_counts(self.cnt.data_ptr())
                torch.cumsum(self.cnt, dim=0, out=self.offsets[1:])
                edge = self.offsets[self.edge_index].tolist()
                seg = [(int(edge[2 * r]), int(edge[2 * r + 1])) for r in range(world)]  # == segment_bounds(offsets, ranges)
                total = int(edge[-1]) if ranges[-1][1] == n_total else int(self.offsets[-1].item())
                keys = torch.empty((max(total, 1),), dtype=torch.int64, device=dev)
                targets = torch.empty((max(total, 1),), dtype=torch.int32, device=dev)
                _sync()
                b.export_proposals(self.offsets.data_ptr(), keys.data_ptr(), targets.data_ptr())
                recv_k = comm.all_to_all_v([keys[a:z] for (a, z) in seg])
                recv_t = comm.all_to_all_v([targets[a:z] for (a, z) in seg])
                rk = torch.cat([t for i, t in enumerate(recv_k) if i != rank])
                rt = torch.cat([t for i, t in enumerate(recv_t) if i != rank])
                _sync()
                if rk.numel():
                    b.import_proposals(rk.data_ptr(), rt.data_ptr(), rk.numel())
                n_sent = total
            # (4) owner-side merge, (5) global update count for the stop rule (pynndescent_.py:317)
            c = comm.all_reduce_sum(b.descent_merge()) if world > 1 else b.descent_merge()
            info["c"].append(c)
            info["exchanged_records"].append(n_sent)
            info["iters"] = it + 1
            if verbose and rank == 0:
                print("\t", it + 1, " / ", self.n_iters, " c =", c)
            if c <= self.delta * k * n_total:
                break
        _sync()
        b.finalize_device(self.out_idx.data_ptr(), self.out_dist.data_ptr())
        b.synchronize()
        info["stats"] = b.stats()
        return self.out_idx, self.out_dist, info


def sharded_build(comm, x_local, metric="euclidean", n_neighbors=15, n_trees=8, leaf_size=None, max_candidates=None,
                  n_iters=None, delta=0.001, seed=0, max_rptree_depth=200, device_index=None, verbose=False):
    """Convenience wrapper: allocate, build the rows this rank owns, release.

    x_local: torch float32 (n_local, d) tensor on this rank's GPU (its shard of the point set).
    Returns (idx int32 (n_local, k) with GLOBAL neighbour ids, alt-space dist float32 (n_local, k), info dict);
    the tensors stay resident on the GPU."""
    n = torch.tensor([x_local.shape[0]], dtype=torch.int64, device=x_local.device)
    sizes = [int(t.item()) for t in comm.all_gather_v(n)]
    if device_index is None:
        device_index = x_local.device.index if x_local.device.index is not None else torch.cuda.current_device()
    sb = ShardedBuilder(comm, sizes, x_local.shape[1], metric, n_neighbors, n_trees, leaf_size, max_candidates, n_iters,
                        delta, seed, max_rptree_depth, device_index)
    try:
        idx, dist, info = sb.build(x_local, verbose=verbose)
        return idx.clone(), dist.clone(), info
    finally:
        sb.close()

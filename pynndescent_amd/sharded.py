"""Row-sharded multi-GPU index build: one process (or thread) per GPU, exchanges over RCCL.

The reference is single-process; what it offers as a sharding rule is owner-computes over contiguous
vertex ranges (``apply_graph_update_array`` utils.py:709-731, ``new_build_candidates`` utils.py:259-306,
``init_rp_tree`` pynndescent_.py:154-185).  Here the same rule crosses GPUs:

* rank r owns rows ``[lo_r, hi_r)`` of the k-lists; the point set is replicated once (all-gather), so
  candidate vectors never travel again;
* the RP forest is split by TREE: rank r builds ``n_trees/G`` trees over all points, seeds the k-lists
  of every point from its own leaves, and the partial lists are shipped to the owners
  (all-to-all-v of k-list row blocks) and merged there;
* per NN-descent iteration every rank scans only ITS OWN rows:
  (1) **threshold all-gather**, 4 bytes per row: the worst distance of every remote row, all the join needs of a
      remote candidate (a stale threshold only admits extra proposals);
  (2) **reverse-offer all-to-all-v** (exchange X2): an edge ``v -> u`` whose target ``u`` lives on another rank becomes a
      12-byte record ``(u | class, priority | v)`` for u's owner, which folds it into its slot banks exactly like a
      local offer -- the cross-process form of the ownership test ``utils.py:266-273``;
  (3) local join of the owned vertices; a proposal for a remote target skips the membership test (the owner
      dedups in its merge, ``utils.py:489-492``);
  (4) **proposal all-to-all-v** of ``(target, dist | source)`` records, owner-side merge (``utils.py:721-731``);
  (5) **all-reduce** of the update count for the stop rule (pynndescent_.py:317).

The device-side halves are C-ABI entry points of ``include/pynnd_amd.h``; they are stream-ordered and the handle runs
on this module's torch stream (``nnd_set_stream``), so kernels and collectives need no host synchronisation between
them: the host waits only where it needs a number (record counts, the update count).  This module is host plumbing
(torch tensors as device buffers, ``torch.distributed`` -- backend "nccl" is RCCL on ROCm).
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

    def all_to_all_v(self, send, rcounts=None, return_counts=False):
        """send[s] goes to rank s (first-dimension sizes arbitrary); returns the list received.  ``rcounts``: the
        receive counts when they are known already (a second array with the same segmentation): no count exchange."""
        if self._host_staged() and send[0].is_cuda:
            dev0 = send[0].device
            out = self.all_to_all_v([t.cpu() for t in send], rcounts, True)
            recv = [g.to(dev0) for g in out[0]]
            return (recv, out[1]) if return_counts else recv
        dev = send[0].device
        sc = [int(t.shape[0]) for t in send]
        if rcounts is None:
            counts = torch.tensor(sc, dtype=torch.int64, device=dev)
            rc_t = torch.empty_like(counts)
            self.dist.all_to_all_single(rc_t, counts, group=self.group)
            rc = [int(c) for c in rc_t.tolist()]
        else:
            rc = [int(c) for c in rcounts]
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
            inp = torch.cat([t.reshape((t.shape[0],) + tail) for t in send], dim=0).contiguous()
            out = torch.empty((sum(rc),) + tail, dtype=send[0].dtype, device=dev)
            self.dist.all_to_all_single(out, inp, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
            recv = list(torch.split(out, rc, dim=0))
        return (recv, rc) if return_counts else recv

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

    def _exchange(self, obj, take):
        """Every rank posts ``obj``; ``take(all_posted)`` copies what this rank needs.  The ranks run on different
        streams of one GPU: a rank's writes are complete before it posts, its copies before the buffers are released."""
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        self.s.slots[self.rank] = obj
        self.s.barrier.wait()
        got = take(list(self.s.slots))
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        self.s.barrier.wait()
        return got

    def all_gather_v(self, t):
        return self._exchange(t, lambda posted: [g.clone() for g in posted])

    def all_to_all_v(self, send, rcounts=None, return_counts=False):
        recv = self._exchange(send, lambda posted: [posted[src][self.rank].clone() for src in range(self.world)])
        return (recv, [int(t.shape[0]) for t in recv]) if return_counts else recv

    def all_reduce_sum(self, value):
        return int(sum(self._exchange(int(value), lambda posted: list(posted))))

    def barrier(self):
        self.s.barrier.wait()


# ------------------------------------------------------------------------------------------------

def _sync():
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()


class ShardedBuilder:
    """Persistent state of one rank (HBM allocations survive across builds; bench.py times ``build``)."""

    PROPOSAL_SLOTS_SHIPPED = 24  # region capacity per destination row (of 64 slots); what does not fit travels next iteration

    def __init__(self, comm, shard_sizes, dim, metric="euclidean", n_neighbors=15, n_trees=8, leaf_size=None,
                 max_candidates=None, n_iters=None, delta=0.001, seed=0, max_rptree_depth=200, device_index=None):
        self.comm = comm
        rank, world = comm.rank, comm.world
        sizes = [int(v) for v in shard_sizes]
        assert len(sizes) == world
        self.n_total = sum(sizes)
        bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.bounds = bounds
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
        # the handle runs on THIS builder's torch stream: its kernels and the collectives are ordered by the stream
        with torch.cuda.device(self.dev):
            self.stream = torch.cuda.Stream(device=self.dev)
        self.b.set_stream(self.stream.cuda_stream)
        self.b.set_shard_bounds(bounds, rank)
        self.ks = self.b.row_stride()
        dev = self.dev
        n_own = self.hi - self.lo
        self.own_th = torch.empty((n_own,), dtype=torch.float32, device=dev)
        self.out_idx = torch.empty((n_own, self.k), dtype=torch.int32, device=dev)
        self.out_dist = torch.empty((n_own, self.k), dtype=torch.float32, device=dev)
        # record regions, one per destination rank: reverse offers (at most every owned edge goes to one rank) and proposals
        self.cap_o = max(1, n_own * self.k)
        self.cap_p = max(64, max(z - a for a, z in self.ranges) * self.PROPOSAL_SLOTS_SHIPPED)
        self.off_t = torch.empty((world * self.cap_o,), dtype=torch.int32, device=dev)
        self.off_k = torch.empty((world * self.cap_o,), dtype=torch.int64, device=dev)
        self.prop_t = torch.empty((world * self.cap_p,), dtype=torch.int32, device=dev)
        self.prop_k = torch.empty((world * self.cap_p,), dtype=torch.int64, device=dev)
        self.counts = torch.zeros((world,), dtype=torch.int64, device=dev)
        self._empty_t = torch.empty((0,), dtype=torch.int32, device=dev)
        self._empty_k = torch.empty((0,), dtype=torch.int64, device=dev)

    def close(self):
        self.b.close()

    def build(self, x_local, verbose=False):
        """One complete sharded build.  Returns (idx (n_local,k) GLOBAL ids, alt-space dist, info)."""
        with torch.cuda.device(self.dev):
            self.stream.wait_stream(torch.cuda.current_stream(self.dev))  # x_local was produced on the caller's stream
            with torch.cuda.stream(self.stream):
                out = self._build(x_local, verbose)
                self.stream.synchronize()
        return out

    def _exchange_records(self, buf_t, buf_k, cap, counts):
        """all-to-all-v of the per-destination regions [d*cap, d*cap + counts[d]); returns the records received from the
        OTHER ranks as (targets, keys, n)."""
        comm, rank = self.comm, self.comm.rank
        send_t = [buf_t[d * cap: d * cap + counts[d]] for d in range(comm.world)]
        send_k = [buf_k[d * cap: d * cap + counts[d]] for d in range(comm.world)]
        recv_t, rc = comm.all_to_all_v(send_t, return_counts=True)
        recv_k = comm.all_to_all_v(send_k, rcounts=rc)
        parts_t = [t for i, t in enumerate(recv_t) if i != rank and t.numel()]
        parts_k = [t for i, t in enumerate(recv_k) if i != rank and t.numel()]
        if not parts_t:
            return self._empty_t, self._empty_k, 0
        rt = torch.cat(parts_t) if len(parts_t) > 1 else parts_t[0].contiguous()
        rk = torch.cat(parts_k) if len(parts_k) > 1 else parts_k[0].contiguous()
        return rt, rk, int(rt.numel())

    def _build(self, x_local, verbose):
        comm, b, dev, ks, k = self.comm, self.b, self.dev, self.ks, self.k
        rank, world = comm.rank, comm.world
        lo, hi, ranges, n_total = self.lo, self.hi, self.ranges, self.n_total
        info = {"n_total": n_total, "range": (lo, hi), "local_trees": self.local_trees, "iters": 0, "c": [],
                "exchanged_records": [], "offer_records": [], "proposal_records": []}
        # ---- replicate the point set once (all-gather over xGMI): candidate vectors never travel again ----
        if world > 1:
            x_full = torch.cat(comm.all_gather_v(x_local.contiguous()), dim=0).contiguous()
        else:
            x_full = x_local.contiguous()
        assert x_full.shape == (n_total, self.d)
        b.set_data_device(x_full.data_ptr(), keepalive=x_full)  # prep kernel + k-list reset (stream-ordered)

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
            recv_e, rc = comm.all_to_all_v(send_e, return_counts=True)
            recv_d = comm.all_to_all_v(send_d, rcounts=rc)
            for src in range(world):
                if src != rank and recv_e[src].numel():
                    b.merge_graph_rows(lo, hi, recv_e[src].data_ptr(), recv_d[src].data_ptr())
            del send_e, send_d, recv_e, recv_d
        b.init_random()  # owned rows that are still not full (pynndescent_.py:188-203)

        keep = None
        for it in range(self.n_iters):
            n_off = n_prop = 0
            # (1) thresholds of the remote rows: 4 bytes per row
            if world > 1:
                b.export_thresholds_async(lo, hi, self.own_th.data_ptr())
                all_t = comm.all_gather_v(self.own_th)
                for src, (a, z) in enumerate(ranges):
                    if src != rank and z > a:
                        b.import_thresholds_async(a, z, all_t[src].data_ptr())
            # (2) sampling: own rows only; offers to remote targets travel as records
            b.sample_begin(self.cap_o, self.off_t.data_ptr(), self.off_k.data_ptr(), self.counts.data_ptr())
            if world > 1:
                cnt = [int(c) for c in self.counts.tolist()]  # host wait: the record counts
                assert max(cnt) <= self.cap_o
                rt, rk, n_in = self._exchange_records(self.off_t, self.off_k, self.cap_o, cnt)
                n_off = sum(cnt)
            else:
                rt, rk, n_in = self._empty_t, self._empty_k, 0
            b.sample_finish(rt.data_ptr(), rk.data_ptr(), n_in)
            # (3) local join of the owned vertices
            b.descent_join()
            # (4) proposals for vertices owned elsewhere -> owners
            if world > 1:
                b.proposal_export(self.cap_p, self.prop_t.data_ptr(), self.prop_k.data_ptr(), self.counts.data_ptr())
                cnt = [min(int(c), self.cap_p) for c in self.counts.tolist()]  # host wait
                pt, pk, n_in = self._exchange_records(self.prop_t, self.prop_k, self.cap_p, cnt)
                if n_in:
                    b.import_proposals_async(pk.data_ptr(), pt.data_ptr(), n_in)
                n_prop = sum(cnt)
                keep = (all_t, rt, rk, pt, pk)  # alive until the merge below has drained the stream
            # (5) owner-side merge (reads the counters: host wait), global update count for the stop rule (pynndescent_.py:317)
            c = comm.all_reduce_sum(b.descent_merge()) if world > 1 else b.descent_merge()
            keep = None
            info["c"].append(c)
            info["offer_records"].append(n_off)
            info["proposal_records"].append(n_prop)
            info["exchanged_records"].append(n_off + n_prop)
            info["iters"] = it + 1
            if verbose and rank == 0:
                print("\t", it + 1, " / ", self.n_iters, " c =", c)
            if c <= self.delta * k * n_total:
                break
        del keep
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

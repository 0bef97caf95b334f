"""Row-sharded multi-GPU index build: host plumbing around the C-side shard builder (csrc/shard.hip).

The reference is single-process; what it offers as a sharding rule is owner-computes over contiguous vertex ranges
(``apply_graph_update_array`` utils.py:709-731, ``new_build_candidates`` utils.py:259-306, ``init_rp_tree``
pynndescent_.py:154-185) and ``n_jobs`` (pynndescent_.py:1141-1143).  The whole per-rank build -- point-set all-gather,
forest sharded by cell, k-list row exchange, and per NN-descent iteration the threshold all-gather, the reverse-offer and
proposal all-to-all-v, the update count for the stop rule -- runs INSIDE ``libpynnd_amd.so`` on one HIP stream, with the
exchanges issued from the C side (``ncclGroupStart`` / ``ncclSend`` / ``ncclRecv`` / ``ncclGroupEnd`` over xGMI; scheme in
``include/pynnd_amd.h``).  What is left here:

* ``make_comm``        -- create the rank's communicator: **RCCL** when ``torch.distributed`` runs the ``nccl`` backend
                          (two 128-byte unique ids -- the build's channel and the second channel -- and three status
                          words are all that travels through torch); **HOST** staging with
                          a gloo callback otherwise (two processes sharing one GPU in the tests);
* ``LocalGroup``       -- G ranks as threads of this process on one or several GPUs (LOCAL transport: device copies);
* ``ShardedBuilder``   -- the rank's persistent state (``nnd_shard_t``), ``build(x_local)``;
* ``build_multi``      -- the one-call form, ``nnd_build_multi``: host arrays in and out, one host thread per GPU inside
                          the library -- what ``NNDescent(..., n_devices=G)`` calls.

"""
import ctypes as C

import numpy as np
import torch

from . import _capi


# ------------------------------------------------------------------------------------------------
# partitioning helper (pure python; the C side uses the same formula)

def shard_ranges(n_total, world):
    """Contiguous, near-equal row ranges: rank r owns [n*r//G, n*(r+1)//G)."""
    return [(n_total * r // world, n_total * (r + 1) // world) for r in range(world)]


# ------------------------------------------------------------------------------------------------
# communicators of the C-side shard builder

class Comm:
    """Owns one ``nnd_comm_t``."""

    def __init__(self, handle, world, rank, keep=None):
        self.lib = _capi.load_library()
        self._h = handle
        self.world = world
        self.rank = rank
        self._keep = keep  # callback objects the C side holds pointers to
        self.transport = "local"

    def set_timeout(self, seconds):
        """A rank that waits longer than this for its peers gives up (error return, ncclCommAbort under RCCL)."""
        if self.lib.nnd_comm_set_timeout(self._h, int(seconds * 1000)) != 0:
            raise _capi.NNDError(self.lib.nnd_comm_last_error(None).decode())

    def info(self):
        out = (C.c_int32 * 4)()
        self.lib.nnd_comm_info(self._h, out)
        v = int(out[2])
        return {"transport": {1: "rccl", 2: "local", 3: "host"}.get(int(out[0]), "?"), "ranks": int(out[1]),
                "rccl_version": "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100) if v else None, "second_channel": bool(out[3])}

    def set_serial(self, on=True):
        if self.lib.nnd_comm_local_set_serial(self._h, 1 if on else 0) != 0:
            raise _capi.NNDError(self.lib.nnd_comm_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.nnd_comm_destroy(self._h)
            self._h = _capi._H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LocalGroup:
    """``world`` ranks as threads of this process (LOCAL transport); ``devices[r]`` = HIP ordinal of rank r (default: all
    on device 0 -- several ranks sharing one GPU, which is how the tests run on a one-GPU box)."""

    def __init__(self, world, devices=None):
        lib = _capi.load_library()
        arr = (_capi._H * world)()
        dev = None
        if devices is not None:
            dev = np.ascontiguousarray(devices, np.int32)
            assert dev.shape == (world,)
        if lib.nnd_comm_create_local(arr, world, _capi._ptr(dev)) != 0:
            raise _capi.NNDError(lib.nnd_comm_last_error(None).decode())
        self.comms = [Comm(_capi._H(arr[r]), world, r) for r in range(world)]
        self.devices = [0] * world if devices is None else [int(v) for v in devices]

    def __getitem__(self, r):
        return self.comms[r]

    def close(self):
        for c in self.comms:
            c.close()


def _host_callback(dist, group=None):
    """nnd_host_exchange_fn over a torch.distributed group (gloo): an all-to-all-v of byte segments of host buffers.  The
    transport's barrier is the call with send == recv == NULL (comm.hip): a data exchange in which this rank happens to
    move no byte is NOT a barrier -- with three ranks the other two would sit in point-to-point calls."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)

    def exchange(_user, send, send_off, send_bytes, recv, recv_off, recv_bytes):
        try:
            if not send and not recv:
                dist.barrier(group=group)
                return 0
            ops, keep = [], []
            for peer in range(world):
                sb, rb = int(send_bytes[peer]), int(recv_bytes[peer])
                if peer == rank:
                    if sb:
                        C.memmove(recv + int(recv_off[peer]), send + int(send_off[peer]), sb)
                    continue
                if sb:
                    buf = (C.c_uint8 * sb).from_address(send + int(send_off[peer]))
                    t = torch.frombuffer(buf, dtype=torch.uint8)
                    keep.append(t)
                    ops.append(dist.isend(t, peer, group=group))
                if rb:
                    buf = (C.c_uint8 * rb).from_address(recv + int(recv_off[peer]))
                    t = torch.frombuffer(buf, dtype=torch.uint8)
                    keep.append(t)
                    ops.append(dist.irecv(t, peer, group=group))
            for op in ops:
                op.wait()
            return 0
        except Exception as exc:  # pragma: no cover
            print("pynndescent_amd host exchange failed:", repr(exc))
            return 1

    return _capi.HOST_EXCHANGE_FN(exchange)


def make_comm(device_index, group=None, allow_host_fallback=False, timeout_s=None):
    """``timeout_s``: how long a host wait of a build may go without the peers before this rank gives up (default: the
    library's 120 s).  Expiry is IRREVERSIBLE under RCCL -- the rank calls ``ncclCommAbort`` on both channels, the communicator is
    dead and a new one has to be made -- so under one process per GPU set it no lower than the skew the ranks can have when they
    ENTER a build (data loading, a first-call code-object load): e.g. the timeout of the torch process group.  The HOST transport
    waits inside the caller's callback (gloo) and is bounded by THAT group's timeout, not by this one.

    The rank's communicator under ``torch.distributed``: RCCL when the process group runs the nccl backend -- two unique
    ids (the build's channel and the second channel of the point-set all-gather) are broadcast through torch, the data
    path never touches torch again; HOST staging over gloo when the group itself is gloo (tests).  If the RCCL
    communicator cannot be created on some rank (all ranks agree on that through one all-reduce) the call RAISES on every
    rank; ``allow_host_fallback=True`` falls back to the HOST transport over a gloo group instead -- loudly, and
    ``Comm.transport`` / ``Comm.info()`` say which one is in use: a scaling run must not silently become a staged one."""
    import torch.distributed as dist

    lib = _capi.load_library()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    h = _capi._H()
    if dist.get_backend(group) == "nccl":
        dev = torch.device("cuda", device_index)
        ident = torch.zeros(257, dtype=torch.uint8)  # two ids + rank 0's status (nobody waits in ncclCommInitRank for ids that were never made)
        ok = 1
        why = ""
        if rank == 0:
            buf, buf2 = (C.c_uint8 * 128)(), (C.c_uint8 * 128)()
            if lib.nnd_comm_unique_id(buf) != 0 or lib.nnd_comm_unique_id(buf2) != 0:
                ok, why = 0, lib.nnd_comm_last_error(None).decode()
            ident = torch.tensor(list(buf) + list(buf2) + [ok], dtype=torch.uint8)
        ident = ident.to(dev)
        dist.broadcast(ident, 0, group=group)
        raw = bytes(ident.cpu().tolist())
        if raw[256] != 1 and ok:
            ok, why = 0, "rank 0 could not create the RCCL unique ids"
        if ok and lib.nnd_comm_create_rccl(C.byref(h), raw[:128], world, rank, int(device_index)) != 0:
            ok, why = 0, lib.nnd_comm_last_error(None).decode()
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:  # (the second channel is created only once every rank holds the first: it is collective too)
            if lib.nnd_comm_add_channel_rccl(h, raw[128:]) != 0:
                ok, why = 0, lib.nnd_comm_last_error(None).decode()
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 1:
            c = Comm(h, world, rank)
            c.transport = "rccl"
            if timeout_s is not None:
                c.set_timeout(timeout_s)
            return c
        if h.value:
            lib.nnd_comm_destroy(h)
            h = _capi._H()
        if not allow_host_fallback:
            raise _capi.NNDError("RCCL communicator could not be created on every rank: %s" % (why or "another rank failed"))
        import warnings

        warnings.warn("pynndescent_amd: RCCL communicator unavailable (%s); exchanging through the HOST transport over gloo -- "
                      "correct but slow" % (why or "another rank failed"))
        cb = _host_callback(dist, dist.new_group(backend="gloo"))
    else:
        cb = _host_callback(dist, group)
    if lib.nnd_comm_create_host(C.byref(h), world, rank, int(device_index), C.cast(cb, C.c_void_p), None) != 0:
        raise _capi.NNDError(lib.nnd_comm_last_error(None).decode())
    c = Comm(h, world, rank, keep=cb)
    c.transport = "host"
    if timeout_s is not None:
        c.set_timeout(timeout_s)
    return c


# ------------------------------------------------------------------------------------------------

def _global_params(n_total, dim, metric, n_neighbors, n_trees, leaf_size, max_candidates, n_iters, delta, seed,
                   max_rptree_depth, device):
    """nnd_params of the GLOBAL build with the reference's derived defaults (pynndescent_.py:1009-1012, 1135-1138;
    rp_trees.py:2845) and its RandomState draw order (identical on every rank)."""
    k = int(n_neighbors)
    n_iters = max(5, int(round(np.log2(n_total)))) if n_iters is None else int(n_iters)
    leaf_size = max(60, min(256, 5 * k)) if leaf_size is None else int(leaf_size)
    mc = min(60, k) if max_candidates is None else int(max_candidates)
    rs = np.random.RandomState(seed)
    lim = np.iinfo(np.int32)
    rng_state = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64)
    _search = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64)
    tree_states = rs.randint(lim.min + 1, lim.max - 1, size=(max(int(n_trees), 1), 3)).astype(np.int64)
    metric_code = _capi.METRIC_CODES[metric]
    p = _capi.NNDParams()
    p.n, p.dim, p.metric, p.n_neighbors, p.n_trees, p.leaf_size = int(n_total), int(dim), metric_code, k, int(n_trees), leaf_size
    p.max_depth, p.max_candidates, p.n_iters, p.delta = int(max_rptree_depth), mc, n_iters, float(delta)
    p.device, p.join_blocks = int(device), 0  # (0: the library's choice -- 1 up to 64 neighbours, sub-steps for wider rows)
    for i in range(3):
        p.rng_state[i], p.tree_rng[i] = int(rng_state[i]), int(tree_states[0][i])
    return p


class ShardedBuilder:
    """Persistent state of one rank (HBM allocations survive across builds; bench.py times ``build``)."""

    def __init__(self, comm, shard_sizes, dim, metric="euclidean", n_neighbors=15, n_trees=8, leaf_size=None,
                 max_candidates=None, n_iters=None, delta=0.001, seed=0, max_rptree_depth=200, device_index=None, flags=0):
        self.lib = _capi.load_library()
        self.comm = comm
        sizes = np.ascontiguousarray([int(v) for v in shard_sizes], np.int64)
        assert sizes.shape == (comm.world,)
        self.n_total = int(sizes.sum())
        bounds = np.concatenate([[0], np.cumsum(sizes)])
        self.lo, self.hi = int(bounds[comm.rank]), int(bounds[comm.rank + 1])
        self.k, self.d = int(n_neighbors), int(dim)
        if device_index is None:
            device_index = torch.cuda.current_device()
        self.dev = torch.device("cuda", device_index)
        self.params = _global_params(self.n_total, dim, metric, n_neighbors, n_trees, leaf_size, max_candidates, n_iters, delta,
                                     seed, max_rptree_depth, device_index)
        self.params.flags = int(flags)
        self._h = _capi._H()
        if self.lib.nnd_shard_create(C.byref(self._h), C.byref(self.params), comm._h, _capi._ptr(sizes)) != 0:
            raise _capi.NNDError(self.lib.nnd_shard_last_error(None).decode())
        n_own = self.hi - self.lo
        self.out_idx = torch.empty((n_own, self.k), dtype=torch.int32, device=self.dev)
        self.out_dist = torch.empty((n_own, self.k), dtype=torch.float32, device=self.dev)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.nnd_shard_destroy(self._h)
            self._h = _capi._H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build(self, x_local, verbose=False):
        """One complete sharded build of this rank's rows.  x_local: float32 (n_local, d) tensor on this rank's GPU.
        Returns (idx (n_local, k) GLOBAL ids, alt-space dist, info dict); the tensors stay resident on the GPU."""
        assert x_local.is_cuda and x_local.dtype == torch.float32 and x_local.is_contiguous()
        assert tuple(x_local.shape) == (self.hi - self.lo, self.d)
        cur = torch.cuda.current_stream(self.dev)
        xs = cur.cuda_stream  # the stream that produced x_local: the build waits for it
        if not xs:  # the NULL stream has no handle to wait on, and the shard's stream does not synchronise with it implicitly
            cur.synchronize()
        rc = self.lib.nnd_shard_build(self._h, C.c_void_p(x_local.data_ptr()), C.c_void_p(xs) if xs else None,
                                      C.c_void_p(self.out_idx.data_ptr()), C.c_void_p(self.out_dist.data_ptr()))
        if rc != 0:
            raise _capi.NNDError(self.lib.nnd_shard_last_error(self._h).decode())
        info = self.info()
        info["stats"] = self.stats()
        if verbose and self.comm.rank == 0:
            for it, c in enumerate(info["c"]):
                print("\t", it + 1, " / ", int(self.params.n_iters), " c =", c)
        return self.out_idx, self.out_dist, info

    def info(self):
        inf = _capi.NNDShardInfo()
        self.lib.nnd_shard_get_info(self._h, C.byref(inf))
        return inf.as_dict()

    def stats(self):
        st = _capi.NNDStats()
        self.lib.nnd_shard_get_stats(self._h, C.byref(st))
        return st.as_dict()


def sharded_build(comm, x_local, shard_sizes, metric="euclidean", n_neighbors=15, n_trees=8, leaf_size=None,
                  max_candidates=None, n_iters=None, delta=0.001, seed=0, max_rptree_depth=200, device_index=None,
                  verbose=False):
    """Convenience wrapper: allocate, build the rows this rank owns, release.  Returns (idx int32 (n_local, k) with
    GLOBAL neighbour ids, alt-space dist float32 (n_local, k), info dict); the tensors stay resident on the GPU."""
    if device_index is None:
        device_index = x_local.device.index if x_local.device.index is not None else torch.cuda.current_device()
    sb = ShardedBuilder(comm, shard_sizes, x_local.shape[1], metric, n_neighbors, n_trees, leaf_size, max_candidates, n_iters,
                        delta, seed, max_rptree_depth, device_index)
    try:
        idx, dist, info = sb.build(x_local, verbose=verbose)
        return idx.clone(), dist.clone(), info
    finally:
        sb.close()


def build_multi(x, n_devices, devices=None, metric="euclidean", n_neighbors=15, n_trees=8, leaf_size=None, max_candidates=None,
                n_iters=None, delta=0.001, seed=0, max_rptree_depth=200, rng_state=None, tree_state=None, init_graph=None,
                init_dist=None, old_graph=None):
    """``nnd_build_multi``: the whole build over ``n_devices`` GPUs of this node in ONE call -- host array in, host arrays
    out, one host thread per GPU inside the library, RCCL between distinct GPUs (LOCAL copies when ``devices`` repeats an
    ordinal).  Returns (idx int32 (n, k), alt-space dist float32 (n, k), stats of rank 0, shard info of rank 0).
    ``init_graph`` (n, width) / ``init_dist``: the warm start of ``NNDescent(init_graph=...)`` (``nnd_build_multi_from_graph``:
    no forest, no random fill; every rank seeds its rows from its rows of the graph).  ``old_graph`` = (ids (n, width), alt-space
    distances): the rebuild of ``NNDescent.update()`` (``nnd_build_multi_update``: a fresh forest + the previous graph's surviving
    entries as OLD entries, no random fill)."""
    lib = _capi.load_library()
    x = np.ascontiguousarray(x, np.float32)
    n, d = x.shape
    p = _global_params(n, d, metric, n_neighbors, n_trees, leaf_size, max_candidates, n_iters, delta, seed, max_rptree_depth, 0)
    if rng_state is not None:
        for i in range(3):
            p.rng_state[i] = int(rng_state[i])
    if tree_state is not None:
        for i in range(3):
            p.tree_rng[i] = int(tree_state[i])
    dev = None if devices is None else np.ascontiguousarray(devices, np.int32)
    idx = np.empty((n, int(n_neighbors)), np.int32)
    dist = np.empty((n, int(n_neighbors)), np.float32)
    st, inf = _capi.NNDStats(), _capi.NNDShardInfo()
    err = C.create_string_buffer(1024)
    if old_graph is not None:
        g = np.ascontiguousarray(old_graph[0], np.int32)
        gd = np.ascontiguousarray(old_graph[1], np.float32)
        if g.shape[0] != n or gd.shape != g.shape:
            raise ValueError("the previous graph does not match the data")
        rc = lib.nnd_build_multi_update(C.byref(p), _capi._ptr(x), int(n_devices), _capi._ptr(dev), _capi._ptr(g), _capi._ptr(gd), int(g.shape[1]),
                                        _capi._ptr(idx), _capi._ptr(dist), C.byref(st), C.byref(inf), err, 1024)
    elif init_graph is not None:
        g = np.ascontiguousarray(init_graph, np.int32)
        gd = None if init_dist is None else np.ascontiguousarray(init_dist, np.float32)
        if g.shape[0] != n or (gd is not None and gd.shape != g.shape):
            raise ValueError("init graph / distances do not match the data")
        rc = lib.nnd_build_multi_from_graph(C.byref(p), _capi._ptr(x), int(n_devices), _capi._ptr(dev), _capi._ptr(g), _capi._ptr(gd),
                                            int(g.shape[1]), _capi._ptr(idx), _capi._ptr(dist), C.byref(st), C.byref(inf), err, 1024)
    else:
        rc = lib.nnd_build_multi(C.byref(p), _capi._ptr(x), int(n_devices), _capi._ptr(dev), _capi._ptr(idx), _capi._ptr(dist),
                                 C.byref(st), C.byref(inf), err, 1024)
    if rc != 0:
        raise _capi.NNDError(err.value.decode())
    return idx, dist, st.as_dict(), inf.as_dict()

"""ctypes binding of libpynnd_amd.so (include/pynnd_amd.h).

This is the whole FFI: plain pointers and sizes, no torch types.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C pynndescent_amd/csrc``; if it is missing or no gfx950 device
is present the build path fails loudly -- there is no CPU fallback."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYNND_AMD_LIB", os.path.join(_HERE, "libpynnd_amd.so"))  # override: A/B experiments only

NND_METRIC_SQEUCLIDEAN = 0
NND_METRIC_ALT_COSINE = 1
# the reference's metric names on this path -> the space the kernels work in ("sqeuclidean" is that space itself: no correction,
# distances.py named_distances / fast_distance_alternatives)
METRIC_CODES = {"euclidean": NND_METRIC_SQEUCLIDEAN, "l2": NND_METRIC_SQEUCLIDEAN, "sqeuclidean": NND_METRIC_SQEUCLIDEAN,
                "cosine": NND_METRIC_ALT_COSINE}
NND_FLAG_NO_GRAPH = 1  # auxiliary handle: no k-lists / candidate / proposal tables (pruning pass, hub tree)
NND_FLAG_NO_PREP = 2   # ... and no prepared copy of the rows (hub tree only)
NND_FLAG_TEST_SELECT_WAVE = 4  # test hook: the one-wave-per-vertex selection kernel
NND_FLAG_TEST_SMALL_REGIONS = 8  # test hook (sharded build): tiny proposal regions, so that records are deferred
NND_FLAG_TEST_ROUTE_PLAIN = 16  # test hook: the forest's routing pass as one walk per (tree, point) through global memory
NND_FLAG_TEST_FOREST_BY_TREE = 32  # test hook (sharded build): forest split by tree where it would be sharded by cell
NND_FLAG_TEST_FAIL = 64  # test hook (sharded build): this rank returns an error at the start of its second iteration
NND_FLAG_TEST_VANISH = 128  # ... or returns there without telling anybody (a killed process)
NND_FLAG_TEST_FOREST_FALLBACK_TOPS = 512  # test hook (sharded build): a rank reports that the by-cell forest cannot be built (at the tops)
NND_FLAG_TEST_FOREST_FALLBACK_SHARE = 1024  # ... at the owners' shares; both flags: at the over-long cells
NND_FLAG_TEST_GATHER_INLINE = 8192  # test hook (sharded build): the id / threshold gather on the build's channel instead of the second one
NND_FLAG_TEST_JOIN_UNSTAGED = 16384  # test hook: k_local_join_w with the membership lists read from global memory instead of staged in LDS
NND_FLAG_TEST_SELECT_HALF = 4096  # test hook: the fused selection with 32 lanes per vertex where 16 would do (k <= 16, max_candidates <= 16)
NND_FLAG_TEST_SAMPLE_NOMEM = 2048  # test hook: the sampler's record regions "cannot be allocated": the handle must fall back to the hashed slots
NND_FLAG_TEST_SAMPLE_ATOMIC = 256  # test hook: reverse offers by one global atomicMin per edge (rounds 1-4) instead of the bucketed transposition


class NNDParams(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("dim", C.c_int32),
        ("metric", C.c_int32),
        ("n_neighbors", C.c_int32),
        ("n_trees", C.c_int32),
        ("leaf_size", C.c_int32),
        ("max_depth", C.c_int32),
        ("max_candidates", C.c_int32),
        ("n_iters", C.c_int32),
        ("delta", C.c_float),
        ("rng_state", C.c_int64 * 3),
        ("tree_rng", C.c_int64 * 3),
        ("device", C.c_int32),
        ("join_blocks", C.c_int32),
        ("flags", C.c_int32),
        ("reserved", C.c_int32 * 5),
    ]


class NNDSearchGraphStats(C.Structure):
    _fields_ = [("forward_nnz", C.c_int64), ("reverse_nnz", C.c_int64), ("union_nnz", C.c_int64), ("final_nnz", C.c_int64),
                ("min_distance", C.c_float), ("max_degree_out", C.c_int32), ("ms_device", C.c_float), ("reserved", C.c_int32 * 3)]


class NNDStats(C.Structure):
    _fields_ = [
        ("n_iters_run", C.c_int64),
        ("n_leaves", C.c_int64),
        ("tree_levels", C.c_int64),
        ("leaf_pairs", C.c_int64),
        ("leaf_rows", C.c_int64),
        ("join_pairs", C.c_int64 * 64),
        ("join_rows", C.c_int64 * 64),
        ("join_active", C.c_int64 * 64),
        ("proposals", C.c_int64 * 64),
        ("updates", C.c_int64 * 64),
        ("ms_prep", C.c_float),
        ("ms_forest", C.c_float),
        ("ms_leaf_init", C.c_float),
        ("ms_random_init", C.c_float),
        ("ms_descent", C.c_float),
        ("ms_finalize", C.c_float),
        ("ms_sample", C.c_float * 64),
        ("ms_join", C.c_float * 64),
        ("ms_merge", C.c_float * 64),
        ("join_mfma", C.c_int64 * 64),
        ("leaf_mfma", C.c_int64),
        ("n_cells", C.c_int64),
        ("join_substeps", C.c_int64 * 64),
    ]

    def as_dict(self):
        it = int(self.n_iters_run)
        m = min(it, 64)
        out = {
            "n_iters_run": it, "n_leaves": int(self.n_leaves), "tree_levels": int(self.tree_levels),
            "leaf_pairs": int(self.leaf_pairs), "leaf_rows": int(self.leaf_rows), "leaf_mfma": int(self.leaf_mfma), "n_cells": int(self.n_cells),
        }
        for name in ("join_pairs", "join_rows", "join_active", "proposals", "updates", "join_mfma", "join_substeps"):
            out[name] = [int(v) for v in getattr(self, name)[:m]]
        for name in ("ms_prep", "ms_forest", "ms_leaf_init", "ms_random_init", "ms_descent", "ms_finalize"):
            out[name] = float(getattr(self, name))
        for name in ("ms_sample", "ms_join", "ms_merge"):
            out[name] = [float(v) for v in getattr(self, name)[:m]]
        return out


class NNDShardInfo(C.Structure):
    _fields_ = [
        ("n_total", C.c_int64), ("own_lo", C.c_int64), ("own_hi", C.c_int64),
        ("world", C.c_int32), ("rank", C.c_int32), ("local_trees", C.c_int32), ("iters", C.c_int32),
        ("c", C.c_int64 * 64),
        ("offer_records", C.c_int64 * 64),
        ("proposal_records", C.c_int64 * 64),
        ("deferred", C.c_int64 * 64),
        ("dropped_offers", C.c_int64),
        ("bytes_sent", C.c_int64),
        ("ms_total", C.c_float), ("ms_allgather", C.c_float), ("ms_klist_exchange", C.c_float),
        ("n_sections", C.c_int32),
        ("section_ms", C.c_float * 256),
        ("section_bytes", C.c_int64 * 256),
        ("forest_by_cell", C.c_int32),
        ("n_sections_overlap", C.c_int32),
        ("forest_positions", C.c_int64),
        ("gather_bytes", C.c_int64 * 64),
        ("gather_section", C.c_int32 * 64),
    ]

    def as_dict(self):
        it = min(int(self.iters), 64)
        ns = int(self.n_sections)
        return {"n_total": int(self.n_total), "range": (int(self.own_lo), int(self.own_hi)), "world": int(self.world),
                "rank": int(self.rank), "local_trees": int(self.local_trees), "iters": int(self.iters),
                "c": [int(v) for v in self.c[:it]], "offer_records": [int(v) for v in self.offer_records[:it]],
                "proposal_records": [int(v) for v in self.proposal_records[:it]],
                "exchanged_records": [int(a) + int(b) for a, b in zip(self.offer_records[:it], self.proposal_records[:it])],
                "deferred": [int(v) for v in self.deferred[:it]], "dropped_offers": int(self.dropped_offers),
                "bytes_sent": int(self.bytes_sent), "ms_total": float(self.ms_total),
                "ms_allgather": float(self.ms_allgather), "ms_klist_exchange": float(self.ms_klist_exchange),
                "section_ms": [float(v) for v in self.section_ms[:ns]],
                "section_bytes": [int(v) for v in self.section_bytes[:ns]],
                "forest_by_cell": bool(self.forest_by_cell), "n_sections_overlap": int(self.n_sections_overlap),
                "forest_positions": int(self.forest_positions),
                "gather_bytes": [int(v) for v in self.gather_bytes[:it]], "gather_section": [int(v) for v in self.gather_section[:it]]}


HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p,
                               C.POINTER(C.c_int64), C.POINTER(C.c_int64))


class NNDPruneOpts(C.Structure):
    _fields_ = [
        ("prune_probability", C.c_float),
        ("degree_aware", C.c_int32),
        ("max_degree", C.c_int32),
        ("aggressiveness", C.c_float),
        ("alpha", C.c_float),
        ("seed", C.c_uint32),
        ("reserved", C.c_int32 * 2),
    ]


# every symbol include/pynnd_amd.h declares: (name, restype, argtypes)
_H = C.c_void_p
_SIGNATURES = [
    ("nnd_abi_version", C.c_int32, []),
    ("nnd_last_global_error", C.c_char_p, []),
    ("nnd_last_error", C.c_char_p, [_H]),
    ("nnd_create", C.c_int32, [C.POINTER(_H), C.POINTER(NNDParams)]),
    ("nnd_destroy", C.c_int32, [_H]),
    ("nnd_set_data_host", C.c_int32, [_H, C.c_void_p]),
    ("nnd_set_data_device", C.c_int32, [_H, C.c_void_p]),
    ("nnd_data_nonfinite", C.c_int32, [_H, C.POINTER(C.c_int32)]),
    ("nnd_release_pending", C.c_int32, []),
    ("nnd_host_copy", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("nnd_host_sqrt_f32", C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64]),
    ("nnd_host_alloc", C.c_void_p, [C.c_int64]),
    ("nnd_host_free", C.c_int32, [C.c_void_p]),
    ("nnd_make_forest", C.c_int32, [_H]),
    ("nnd_leaf_array_shape", C.c_int32, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    ("nnd_get_leaf_array", C.c_int32, [_H, C.c_void_p]),
    ("nnd_reset_graph", C.c_int32, [_H]),
    ("nnd_init_from_leaves", C.c_int32, [_H]),
    ("nnd_init_from_leaf_array", C.c_int32, [_H, C.c_void_p, C.c_int64, C.c_int32]),
    ("nnd_init_random", C.c_int32, [_H]),
    ("nnd_init_from_graph", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_int32]),
    ("nnd_init_from_neighbor_graph", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_int32]),
    ("nnd_descent_iter", C.c_int32, [_H, C.POINTER(C.c_int64)]),
    ("nnd_descent", C.c_int32, [_H]),
    ("nnd_finalize_host", C.c_int32, [_H, C.c_void_p, C.c_void_p]),
    ("nnd_finalize_device", C.c_int32, [_H, C.c_void_p, C.c_void_p]),
    ("nnd_build_device", C.c_int32, [_H, C.c_void_p, C.c_void_p]),
    ("nnd_build", C.c_int32, [C.POINTER(NNDParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                              C.c_void_p, C.POINTER(NNDStats), C.c_char_p, C.c_int32]),
    ("nnd_get_stats", C.c_int32, [_H, C.POINTER(NNDStats)]),
    ("nnd_synchronize", C.c_int32, [_H]),
    ("nnd_get_graph", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nnd_get_candidates", C.c_int32, [_H, C.c_void_p, C.c_void_p]),
    ("nnd_sample_candidates", C.c_int32, [_H]),
    ("nnd_pairwise_gram", C.c_int32, [_H, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    ("nnd_descent_sample", C.c_int32, [_H]),
    ("nnd_descent_join", C.c_int32, [_H]),
    ("nnd_set_stream", C.c_int32, [_H, C.c_void_p]),
    ("nnd_comm_unique_id", C.c_int32, [C.c_void_p]),
    ("nnd_comm_create_rccl", C.c_int32, [C.POINTER(_H), C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    ("nnd_comm_create_local", C.c_int32, [C.POINTER(_H), C.c_int32, C.c_void_p]),
    ("nnd_comm_create_host", C.c_int32, [C.POINTER(_H), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    ("nnd_comm_add_channel_rccl", C.c_int32, [_H, C.c_void_p]),
    ("nnd_comm_set_timeout", C.c_int32, [_H, C.c_int64]),
    ("nnd_comm_info", C.c_int32, [_H, C.POINTER(C.c_int32)]),
    ("nnd_comm_self_test", C.c_int32, [_H]),
    ("nnd_comm_destroy", C.c_int32, [_H]),
    ("nnd_comm_abort", C.c_int32, [_H]),
    ("nnd_comm_local_set_serial", C.c_int32, [_H, C.c_int32]),
    ("nnd_comm_last_error", C.c_char_p, [_H]),
    ("nnd_shard_create", C.c_int32, [C.POINTER(_H), C.POINTER(NNDParams), _H, C.c_void_p]),
    ("nnd_shard_build", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("nnd_shard_get_info", C.c_int32, [_H, C.POINTER(NNDShardInfo)]),
    ("nnd_shard_get_stats", C.c_int32, [_H, C.POINTER(NNDStats)]),
    ("nnd_shard_handle", _H, [_H]),
    ("nnd_shard_destroy", C.c_int32, [_H]),
    ("nnd_shard_last_error", C.c_char_p, [_H]),
    ("nnd_build_multi", C.c_int32, [C.POINTER(NNDParams), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(NNDStats), C.POINTER(NNDShardInfo), C.c_char_p, C.c_int32]),
    ("nnd_build_multi_from_graph", C.c_int32, [C.POINTER(NNDParams), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                               C.c_void_p, C.c_void_p, C.POINTER(NNDStats), C.POINTER(NNDShardInfo), C.c_char_p, C.c_int32]),
    ("nnd_shard_build_from_graph", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    ("nnd_shard_build_update", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    ("nnd_build_multi_update", C.c_int32, [C.POINTER(NNDParams), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.POINTER(NNDStats), C.POINTER(NNDShardInfo), C.c_char_p, C.c_int32]),
    ("nnd_diversify_host", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.POINTER(NNDPruneOpts), C.c_void_p]),
    ("nnd_diversify_csr_host", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(NNDPruneOpts),
                                           C.c_void_p]),
    ("nnd_degree_prune_host", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]),
    ("nnd_search_graph", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_uint32,
                                     C.c_void_p, C.c_void_p, C.POINTER(NNDSearchGraphStats)]),
    ("nnd_search_graph_fetch", C.c_int32, [_H, C.c_void_p, C.c_void_p]),
    ("nnd_hub_tree_build", C.c_int32, [_H, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    ("nnd_hub_tree_fetch", C.c_int32, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    ("nnd_searcher_create", C.c_int32, [C.POINTER(_H), C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_int32,
                                        C.c_void_p]),
    ("nnd_searcher_query", C.c_int32, [_H, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    ("nnd_searcher_last_spilled", C.c_int64, [_H]),
    ("nnd_searcher_set_tier", C.c_int32, [_H, C.c_int32]),
    ("nnd_searcher_destroy", C.c_int32, [_H]),
    ("nnd_searcher_last_error", C.c_char_p, [_H]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


def load_library():
    """dlopen libpynnd_amd.so and bind every entry point.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C pynndescent_amd/csrc`. pynndescent_amd has no CPU fallback." % LIB_PATH
        )
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in _SIGNATURES:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class NNDError(RuntimeError):
    pass


class Builder:
    """Thin object wrapper over an ``nnd_handle_t`` (one GPU, one stream)."""

    def __init__(self, n, dim, metric, n_neighbors, n_trees, leaf_size, max_depth, max_candidates, n_iters, delta,
                 rng_state, tree_rng, device=0, join_blocks=0, flags=0):
        self.lib = load_library()
        p = NNDParams()
        p.n, p.dim, p.metric = int(n), int(dim), int(metric)
        p.n_neighbors, p.n_trees, p.leaf_size = int(n_neighbors), int(n_trees), int(leaf_size)
        p.max_depth, p.max_candidates, p.n_iters = int(max_depth), int(max_candidates), int(n_iters)
        p.delta = float(delta)
        for i in range(3):
            p.rng_state[i] = int(rng_state[i])
            p.tree_rng[i] = int(tree_rng[i])
        p.device, p.join_blocks, p.flags = int(device), int(join_blocks), int(flags)
        self.params = p
        self.n, self.dim, self.k, self.mc = int(n), int(dim), int(n_neighbors), int(max_candidates)
        self._h = _H()
        if self.lib.nnd_create(C.byref(self._h), C.byref(p)) != 0:
            raise NNDError(self.lib.nnd_last_global_error().decode())
        self._keepalive = None

    def _check(self, rc):
        if rc != 0:
            raise NNDError(self.lib.nnd_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.nnd_destroy(self._h)
            self._h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data -----------------------------------------------------------------------------------
    def set_data_host(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape == (self.n, self.dim)
        self._check(self.lib.nnd_set_data_host(self._h, _ptr(x)))

    def data_nonfinite(self):
        """True when the point set handed in held a NaN or an infinity (flag raised by the prep kernel)."""
        out = C.c_int32()
        self._check(self.lib.nnd_data_nonfinite(self._h, C.byref(out)))
        return bool(out.value)

    def set_data_device(self, dev_ptr, keepalive=None):
        """dev_ptr: integer address of a float32 (n, dim) C-contiguous device buffer (e.g. tensor.data_ptr()).

        The prep kernel reads the buffer right away on the handle's stream: the producer must be done
        (``torch.cuda.synchronize()`` / stream sync) unless the handle runs on the producing stream (``set_stream``)."""
        self._keepalive = keepalive
        self._check(self.lib.nnd_set_data_device(self._h, C.c_void_p(int(dev_ptr))))

    # -- stages ---------------------------------------------------------------------------------
    def make_forest(self):
        self._check(self.lib.nnd_make_forest(self._h))

    def leaf_array(self):
        nl, ms = C.c_int64(), C.c_int32()
        self._check(self.lib.nnd_leaf_array_shape(self._h, C.byref(nl), C.byref(ms)))
        out = np.empty((max(nl.value, 1), max(ms.value, 1)), np.int32)
        self._check(self.lib.nnd_get_leaf_array(self._h, _ptr(out)))
        return out[: nl.value]

    def reset_graph(self):
        self._check(self.lib.nnd_reset_graph(self._h))

    def init_from_leaves(self):
        self._check(self.lib.nnd_init_from_leaves(self._h))

    def init_from_leaf_array(self, leaf_array):
        """init_rp_tree on the caller's leaves: int32 (n_leaves, max_leaf_size), -1 padded (rptree_leaf_array's table)."""
        la = np.ascontiguousarray(leaf_array, dtype=np.int32)
        assert la.ndim == 2
        self._check(self.lib.nnd_init_from_leaf_array(self._h, _ptr(la), la.shape[0], la.shape[1]))

    def init_random(self):
        self._check(self.lib.nnd_init_random(self._h))

    def init_from_graph(self, idx, dist=None):
        idx = np.ascontiguousarray(idx, np.int32)
        dist = None if dist is None else np.ascontiguousarray(dist, np.float32)
        self._check(self.lib.nnd_init_from_graph(self._h, _ptr(idx), _ptr(dist), idx.shape[1]))

    def init_from_neighbor_graph(self, idx, dist):
        idx = np.ascontiguousarray(idx, np.int32)
        dist = np.ascontiguousarray(dist, np.float32)
        self._check(self.lib.nnd_init_from_neighbor_graph(self._h, _ptr(idx), _ptr(dist), idx.shape[1]))

    def descent_iter(self):
        c = C.c_int64()
        self._check(self.lib.nnd_descent_iter(self._h, C.byref(c)))
        return c.value

    def descent(self):
        self._check(self.lib.nnd_descent(self._h))

    def sample_candidates(self):
        self._check(self.lib.nnd_sample_candidates(self._h))

    def finalize(self):
        idx = host_pool.empty((self.n, self.k), np.int32)
        dist = host_pool.empty((self.n, self.k), np.float32)
        self._check(self.lib.nnd_finalize_host(self._h, _ptr(idx), _ptr(dist)))
        return idx, dist

    def finalize_device(self, idx_ptr, dist_ptr):
        self._check(self.lib.nnd_finalize_device(self._h, C.c_void_p(int(idx_ptr)), C.c_void_p(int(dist_ptr))))

    def build_device(self, idx_ptr, dist_ptr):
        self._check(self.lib.nnd_build_device(self._h, C.c_void_p(int(idx_ptr)), C.c_void_p(int(dist_ptr))))

    def synchronize(self):
        self._check(self.lib.nnd_synchronize(self._h))

    # -- introspection --------------------------------------------------------------------------
    def stats(self, raw=False):
        """The handle's statistics block; ``raw``: the ctypes structure itself (per-iteration arrays beyond ``n_iters_run`` too:
        ``descent_join`` leaves the counters of a join that no finished iteration owns yet at index ``n_iters_run``)."""
        s = NNDStats()
        self._check(self.lib.nnd_get_stats(self._h, C.byref(s)))
        return s if raw else s.as_dict()

    def graph(self):
        idx = np.empty((self.n, self.k), np.int32)
        dist = np.empty((self.n, self.k), np.float32)
        flags = np.empty((self.n, self.k), np.uint8)
        self._check(self.lib.nnd_get_graph(self._h, _ptr(idx), _ptr(dist), _ptr(flags)))
        return idx, dist, flags

    def candidates(self):
        new = np.empty((self.n, self.mc), np.int32)
        old = np.empty((self.n, self.mc), np.int32)
        self._check(self.lib.nnd_get_candidates(self._h, _ptr(new), _ptr(old)))
        return new, old

    def set_stream(self, stream_ptr):
        """Run the handle on the caller's HIP stream (integer address; 0 / None: the handle's own stream again)."""
        self._check(self.lib.nnd_set_stream(self._h, C.c_void_p(int(stream_ptr)) if stream_ptr else None))

    def descent_sample(self):
        self._check(self.lib.nnd_descent_sample(self._h))

    def descent_join(self):
        self._check(self.lib.nnd_descent_join(self._h))

    @staticmethod
    def _prune_opts(prune_probability=1.0, degree_aware=False, max_degree=1, aggressiveness=1.0, alpha=1.0, seed=0):
        o = NNDPruneOpts()
        o.prune_probability, o.degree_aware, o.max_degree = float(prune_probability), int(bool(degree_aware)), int(max_degree)
        o.aggressiveness, o.alpha, o.seed = float(aggressiveness), float(alpha), int(seed) & 0xFFFFFFFF
        return o

    def diversify(self, idx, dist, degree=None, **opts):
        """diversify / diversify_degree_aware; opts: prune_probability, degree_aware, max_degree, aggressiveness, alpha, seed."""
        idx = np.ascontiguousarray(idx, np.int32).copy()
        dist = np.ascontiguousarray(dist, np.float32).copy()
        assert idx.shape == (self.n, self.k) and dist.shape == idx.shape
        o = self._prune_opts(**opts)
        degree = None if degree is None else np.ascontiguousarray(degree, np.int32)
        self._check(self.lib.nnd_diversify_host(self._h, _ptr(idx), _ptr(dist), C.byref(o), _ptr(degree)))
        return idx, dist

    def diversify_csr(self, indptr, indices, data, degree=None, **opts):
        indptr = np.ascontiguousarray(indptr, np.int32)
        indices = np.ascontiguousarray(indices, np.int32)
        data = np.ascontiguousarray(data, np.float32).copy()
        assert indptr.shape[0] == self.n + 1
        o = self._prune_opts(**opts)
        degree = None if degree is None else np.ascontiguousarray(degree, np.int32)
        self._check(self.lib.nnd_diversify_csr_host(self._h, _ptr(indptr), _ptr(indices), _ptr(data), data.shape[0],
                                                    C.byref(o), _ptr(degree)))
        return data

    def search_graph(self, idx, dist, n_neighbors, pruning_degree_multiplier=1.5, diversify_prob=1.0, degree_aware=False,
                     degree_prune_aggressiveness=1.0, seed=0, on_device=False, want_forward=False):
        """The whole pruning pass on the device (csrc/searchgraph.hip).  idx / dist: numpy (n, k) arrays, or device addresses
        (ints) when ``on_device``.  Returns (indptr, indices, stats dict[, forward rows, forward distances])."""
        st = NNDSearchGraphStats()
        fr = fd = None
        if want_forward:
            fr = np.empty((self.n, self.k), np.int32)
            fd = np.empty((self.n, self.k), np.float32)
        if on_device:
            pi, pd = C.c_void_p(int(idx)), C.c_void_p(int(dist))
        else:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            dist = np.ascontiguousarray(dist, dtype=np.float32)
            assert idx.shape == (self.n, self.k) and dist.shape == (self.n, self.k)
            pi, pd = _ptr(idx), _ptr(dist)
        self._check(self.lib.nnd_search_graph(self._h, pi, pd, 1 if on_device else 0, int(n_neighbors), float(pruning_degree_multiplier),
                                              float(diversify_prob), 1 if degree_aware else 0, float(degree_prune_aggressiveness),
                                              int(seed) & 0xFFFFFFFF, _ptr(fr) if want_forward else None, _ptr(fd) if want_forward else None,
                                              C.byref(st)))
        indptr = np.empty(self.n + 1, np.int32)
        indices = np.empty(max(int(st.final_nnz), 1), np.int32)
        self._check(self.lib.nnd_search_graph_fetch(self._h, _ptr(indptr), _ptr(indices)))
        stats = {f: getattr(st, f) for f, _ in NNDSearchGraphStats._fields_ if f != "reserved"}
        out = (indptr, indices[: int(st.final_nnz)], stats)
        return out + (fr, fd) if want_forward else out

    def degree_prune(self, indptr, data, max_degree):
        indptr = np.ascontiguousarray(indptr, np.int32)
        data = np.ascontiguousarray(data, np.float32).copy()
        self._check(self.lib.nnd_degree_prune_host(self._h, _ptr(indptr), _ptr(data), data.shape[0], int(max_degree)))
        return data

    def hub_tree(self, rank_order, leaf_size, max_depth):
        """make_hub_tree + convert_tree_format on the device; returns (hyperplanes, offsets, children, indices, leaf_size)."""
        ro = np.ascontiguousarray(rank_order, np.int32)
        assert ro.shape == (self.n,)
        nn = C.c_int64()
        self._check(self.lib.nnd_hub_tree_build(self._h, _ptr(ro), int(leaf_size), int(max_depth), C.byref(nn)))
        hyper = np.empty((nn.value, self.dim), np.float32)
        offs = np.empty((nn.value,), np.float32)
        children = np.empty((nn.value, 2), np.int32)
        indices = np.empty((self.n,), np.int32)
        ml = C.c_int32()
        self._check(self.lib.nnd_hub_tree_fetch(self._h, _ptr(hyper), _ptr(offs), _ptr(children), _ptr(indices), C.byref(ml)))
        return hyper, offs, children, indices, int(ml.value)

    def pairwise_gram(self, rows_a, rows_b):
        a = np.ascontiguousarray(rows_a, np.int32)
        b = np.ascontiguousarray(rows_b, np.int32)
        out = np.empty((a.shape[0], b.shape[0]), np.float32)
        self._check(self.lib.nnd_pairwise_gram(self._h, _ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out)))
        return out


class Searcher:
    """Thin wrapper over an ``nnd_searcher_t``: device copies of a prepared index, batched queries (one wave per query)."""

    def __init__(self, data, search_graph, tree, metric, min_distance, n_neighbors, search_rng_state, device=0):
        self.lib = load_library()
        data = np.ascontiguousarray(data, np.float32)
        self.n, self.dim = data.shape
        indptr = np.ascontiguousarray(search_graph.indptr, np.int32)
        indices = np.ascontiguousarray(search_graph.indices, np.int32)
        if tree is not None:
            hyper = np.ascontiguousarray(tree.hyperplanes, np.float32)
            offs = np.ascontiguousarray(tree.offsets, np.float32)
            children = np.ascontiguousarray(tree.children, np.int32)
            tidx = np.ascontiguousarray(tree.indices, np.int32)
            n_nodes = hyper.shape[0]
        else:
            hyper = offs = children = tidx = None
            n_nodes = 0
        rng = np.ascontiguousarray(search_rng_state, np.int64)
        self._h = _H()
        rc = self.lib.nnd_searcher_create(C.byref(self._h), int(device), self.n, self.dim, int(metric), _ptr(data), _ptr(indptr),
                                          _ptr(indices), int(indices.shape[0]), _ptr(hyper), _ptr(offs), _ptr(children), _ptr(tidx),
                                          int(n_nodes), float(min_distance), int(n_neighbors), _ptr(rng))
        if rc != 0:
            raise NNDError(self.lib.nnd_searcher_last_error(None).decode())

    def query(self, queries, k, epsilon):
        q = np.ascontiguousarray(queries, np.float32)
        assert q.ndim == 2 and q.shape[1] == self.dim
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.float32)
        if self.lib.nnd_searcher_query(self._h, _ptr(q), q.shape[0], int(k), float(epsilon), _ptr(idx), _ptr(dist)) != 0:
            raise NNDError(self.lib.nnd_searcher_last_error(self._h).decode())
        return idx, dist

    def last_spilled(self):
        """Queries of the last call whose search outgrew the LDS structures and ran on the global-memory tier."""
        return int(self.lib.nnd_searcher_last_spilled(self._h))

    def set_tier(self, tier):
        """0: automatic (LDS tier, overflowing queries re-run on the global-memory tier); 1: every query on the latter."""
        if self.lib.nnd_searcher_set_tier(self._h, int(tier)) != 0:
            raise NNDError(self.lib.nnd_searcher_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.nnd_searcher_destroy(self._h)
            self._h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostPool:
    """Result arrays in pinned host memory, recycled (round 6; include/pynnd_amd.h nnd_host_alloc).

    ``empty(shape, dtype)`` is ``numpy.empty`` for arrays of 4 MB and more, backed by a pinned buffer: the pages are resident (a
    fresh 60 MB numpy array costs ~15 k page faults on first touch: 12 of the 57 ms of ``NNDescent(x).neighbor_graph`` at
    1 M x 15) and the library copies a finished graph into it with one DMA.  A buffer returns to the pool when the last numpy
    view of it is garbage-collected and serves the next request of the same size: a process that builds index after index
    allocates its result arrays once.  Without a device, or beyond ``max_bytes`` of pinned memory, ``numpy.empty``."""

    def __init__(self, min_bytes=4 << 20, max_bytes=4 << 30, keep_per_size=6):
        import threading

        self.min_bytes, self.max_bytes, self.keep = min_bytes, max_bytes, keep_per_size
        self._free, self._held, self._lock = {}, 0, threading.Lock()

    def _release(self, ptr, nbytes):
        with self._lock:
            lst = self._free.setdefault(nbytes, [])
            if len(lst) < self.keep:
                lst.append(ptr)
                return
            self._held -= nbytes
        try:
            load_library().nnd_host_free(C.c_void_p(ptr))
        except Exception:  # (interpreter shutdown)
            pass

    def empty(self, shape, dtype):
        import weakref

        dtype = np.dtype(dtype)
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes < self.min_bytes:
            return np.empty(shape, dtype)
        ptr = None
        with self._lock:
            lst = self._free.get(nbytes)
            if lst:
                ptr = lst.pop()
            elif self._held + nbytes <= self.max_bytes:
                self._held += nbytes
                ptr = 0
        if ptr == 0:
            ptr = load_library().nnd_host_alloc(nbytes)
            if not ptr:
                with self._lock:
                    self._held -= nbytes
                ptr = None
        if ptr is None:
            return np.empty(shape, dtype)
        buf = (C.c_char * nbytes).from_address(ptr)
        fin = weakref.finalize(buf, self._release, ptr, nbytes)
        fin.atexit = False  # (process exit releases the memory)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def trim(self):
        """Give the idle buffers back."""
        with self._lock:
            items = [(p, nb) for nb, lst in self._free.items() for p in lst]
            self._free.clear()
            self._held -= sum(nb for _, nb in items)
        for p, _ in items:
            load_library().nnd_host_free(C.c_void_p(p))


host_pool = HostPool()


def host_copy(a):
    """``a.copy()`` for a C-contiguous array, the pages of the fresh destination touched by several host threads."""
    if not a.flags.c_contiguous or a.nbytes < (4 << 20):
        return a.copy()
    out = host_pool.empty(a.shape, a.dtype)
    if load_library().nnd_host_copy(_ptr(out), _ptr(a), a.nbytes) != 0:
        raise NNDError(load_library().nnd_last_global_error().decode())
    return out


def host_sqrt(a):
    """``numpy.sqrt(a)`` for a C-contiguous float32 array (IEEE sqrtf per element: the same bits), several host threads."""
    if a.dtype != np.float32 or not a.flags.c_contiguous or a.nbytes < (4 << 20):
        return np.sqrt(a)
    out = host_pool.empty(a.shape, a.dtype)
    if load_library().nnd_host_sqrt_f32(_ptr(out), _ptr(a), a.size) != 0:
        raise NNDError(load_library().nnd_last_global_error().decode())
    return out

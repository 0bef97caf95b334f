"""``NNDescent`` -- drop-in for the BUILD path of ``pynndescent.NNDescent`` on an MI355X.

Mirrors the reference constructor (pynndescent/pynndescent_.py:976-1269): same keyword arguments in
the same positional order, same derived defaults, same RandomState draw order, same attributes after
construction, same errors and warnings.  The dense euclidean / cosine branch
(pynndescent_.py:1221-1260) -- ``make_forest`` + ``rptree_leaf_array`` + ``nn_descent`` -- runs on
the GPU through the C ABI of ``include/pynnd_amd.h``.  Nothing here computes neighbours on the CPU:
if the HIP library or a gfx950 device is missing, construction raises.

Also on the GPU: ``update`` (warm start, pynndescent_.py:2381-2553), ``build_search_graph`` (the pruning
pass of ``_init_search_graph``, all diversify methods), ``prepare`` (hub search tree + reordering) and
``query``.  Out of scope: sparse input, metrics other than euclidean / l2 / sqeuclidean / cosine, ``n_neighbors`` above 256 or
``max_candidates`` above 128 (``query``: more than 256 results per query).  Those raise ``NotImplementedError`` naming the reference entry point to use
instead; ``pynndescent_amd.make_index`` hands such inputs to ``pynndescent.NNDescent`` when it is importable.
"""
import time
from warnings import warn

import numpy as np
from sklearn.utils import check_array, check_random_state

from . import _capi

INT32_MIN = np.iinfo(np.int32).min + 1  # pynndescent_.py:62
INT32_MAX = np.iinfo(np.int32).max - 1  # pynndescent_.py:63

_METRIC_CODES = _capi.METRIC_CODES
# metrics whose trees are angular in the reference (pynndescent_.py:1075-1086)
_ANGULAR_METRICS = ("cosine", "dot", "correlation", "dice", "jaccard", "hellinger", "hamming", "bit_hamming",
                    "bit_jaccard")


def ts():
    """Timestamp used in verbose output (reference utils.py:881-883)."""
    return time.ctime(time.time())


def tau_rand_int(state):
    """Reference utils.py:17-40 on an int64[3] numpy state (used only to warm up search_rng_state)."""
    s = [int(v) for v in state]
    s[0] = (((s[0] & 4294967294) << 12) & 0xFFFFFFFF) ^ ((((s[0] << 13) & 0xFFFFFFFF) ^ s[0]) >> 19)
    s[1] = (((s[1] & 4294967288) << 4) & 0xFFFFFFFF) ^ ((((s[1] << 2) & 0xFFFFFFFF) ^ s[1]) >> 25)
    s[2] = (((s[2] & 4294967280) << 17) & 0xFFFFFFFF) ^ ((((s[2] << 3) & 0xFFFFFFFF) ^ s[2]) >> 11)
    state[0], state[1], state[2] = s
    r = (s[0] ^ s[1] ^ s[2]) & 0xFFFFFFFF
    return r - (1 << 32) if r >= (1 << 31) else r


def correct_alternative_cosine(d):
    """1 - 2^-d (reference distances.py:704-711); float64 like the reference's ufunc."""
    return 1.0 - np.power(2.0, -np.asarray(d, dtype=np.float64))


# (numpy.sqrt; large float32 arrays go through the library's threaded sqrtf -- the same bits, the fresh pages touched in parallel)
# ("sqeuclidean": the reference applies no correction, pynndescent_.py:1271-1298; neighbor_graph still hands out a copy)
_DISTANCE_CORRECTIONS = {"euclidean": _capi.host_sqrt, "l2": _capi.host_sqrt, "sqeuclidean": _capi.host_copy, "cosine": correct_alternative_cosine}


class _DeviceForestSentinel:
    """Stands in for ``_rp_forest``: downstream reference code only null-checks it
    (pynndescent_.py:1353) -- the build consumes nothing but the leaf array."""

    def __init__(self, n_trees, n_leaves, max_leaf_size):
        self.n_trees, self.n_leaves, self.max_leaf_size = n_trees, n_leaves, max_leaf_size

    def __len__(self):
        return self.n_trees


class NNDescent:
    """See ``pynndescent.NNDescent``; constructor signature identical (pynndescent_.py:976-1007)."""

    def __init__(
        self,
        data,
        metric="euclidean",
        metric_kwds=None,
        bit_metric=False,
        n_neighbors=30,
        n_trees=None,
        angular_trees=False,
        leaf_size=None,
        pruning_degree_multiplier=1.5,
        diversify_prob=1.0,
        diversify_method="standard",
        degree_prune_aggressiveness=1.0,
        n_search_trees=1,
        search_tree_leaf_size=None,
        max_search_tree_depth=None,
        quantization=None,
        tree_init=True,
        init_graph=None,
        init_dist=None,
        random_state=None,
        low_memory=True,
        max_candidates=None,
        max_rptree_depth=200,
        n_iters=None,
        delta=0.001,
        n_jobs=None,
        compressed=False,
        parallel_batch_queries=False,
        verbose=False,
        device=0,
        n_devices=1,
        devices=None,
    ):
        """``device``: HIP ordinal of the GPU that builds the index.  ``n_devices`` > 1: the build is row-sharded over that
        many GPUs of this node (``nnd_build_multi``: one host thread per GPU inside the library, RCCL over xGMI; the
        reference's analogue is ``n_jobs``, pynndescent_.py:1141-1143); ``devices`` lists their ordinals (default
        0..n_devices-1; a list that repeats an ordinal puts several ranks on one GPU).  Everything after the build
        (``prepare``, ``query``, ``update``) runs on ``device``."""
        if n_trees is None:
            n_trees = max(3, min(12, int(round(2.0 * np.log10(data.shape[0])))))  # pynndescent_.py:1009-1010
        if n_iters is None:
            n_iters = max(5, int(round(np.log2(data.shape[0]))))  # pynndescent_.py:1011-1012

        self.n_trees = n_trees
        self.angular_trees = angular_trees
        self.n_trees_after_update = max(2, int(np.round(self.n_trees / 3)))
        self.n_neighbors = n_neighbors
        self.metric = metric
        self.metric_kwds = metric_kwds
        self.bit_metric = bit_metric
        self.leaf_size = leaf_size
        self.prune_degree_multiplier = pruning_degree_multiplier
        self.diversify_prob = diversify_prob
        self.diversify_method = diversify_method
        self.degree_prune_aggressiveness = degree_prune_aggressiveness
        self.n_search_trees = n_search_trees
        self.search_tree_leaf_size = search_tree_leaf_size
        self.max_search_tree_depth = max_search_tree_depth
        self.max_rptree_depth = max_rptree_depth
        self.max_candidates = max_candidates
        self.quantization = quantization
        self.low_memory = low_memory
        self.n_iters = n_iters
        self.delta = delta
        self.dim = data.shape[1]
        self.n_jobs = n_jobs
        self.compressed = compressed
        self.parallel_batch_queries = parallel_batch_queries
        self.verbose = verbose
        self.device = device
        self.n_devices = int(n_devices)
        self.devices = None if devices is None else [int(v) for v in devices]
        if self.n_devices < 1 or (self.devices is not None and len(self.devices) != self.n_devices):
            raise ValueError("n_devices must be >= 1 and match len(devices)")

        if callable(metric) or metric not in _METRIC_CODES:
            if callable(metric) or metric in _KNOWN_REFERENCE_METRICS:
                raise NotImplementedError(
                    "pynndescent_amd accelerates the dense euclidean / l2 / sqeuclidean / cosine build only; "
                    "use pynndescent.NNDescent for metric %r" % (metric,)
                )
            raise ValueError("Metric is neither callable, " + "nor a recognised string")  # pynndescent_.py:1292
        try:
            import scipy.sparse

            if scipy.sparse.issparse(data):
                raise NotImplementedError(
                    "sparse input is out of scope for pynndescent_amd; use pynndescent.NNDescent (sparse_nndescent)"
                )
        except ImportError:  # pragma: no cover
            pass

        _check_supported_sizes(n_neighbors, max_candidates, init_graph)
        # pynndescent_.py:1054 check_array(data, dtype=np.float32, order="C") -- minus its single-core scan for NaN / inf
        # (24 ms at 1 M x 128): the prep kernel looks at every value anyway and raises a flag; _raise_if_nonfinite then
        # lets sklearn produce the reference's own error
        data = _check_array_no_scan(data)
        self._input_dtype = np.float32
        self._raw_data = data

        if not tree_init or n_trees == 0 or init_graph is not None:  # pynndescent_.py:1059-1062
            self.tree_init = False
        else:
            self.tree_init = True

        metric_kwds = metric_kwds or {}
        self._dist_args = tuple(metric_kwds.values())
        self.random_state = random_state
        current_random_state = check_random_state(self.random_state)

        self._distance_correction = _DISTANCE_CORRECTIONS[metric]
        self._distance_func = None  # device kernels; see include/pynnd_amd.h NND_METRIC_*
        self._angular_trees = metric in _ANGULAR_METRICS
        self._bit_trees = False
        self._is_sparse = False

        # RandomState draw order of the reference: rng_state, search_rng_state, then the per-tree
        # states inside make_forest (pynndescent_.py:1105-1113, rp_trees.py:2850)
        self.rng_state = current_random_state.randint(INT32_MIN, INT32_MAX, 3).astype(np.int64)
        self.search_rng_state = current_random_state.randint(INT32_MIN, INT32_MAX, 3).astype(np.int64)
        for _ in range(10):
            tau_rand_int(self.search_rng_state)

        n = data.shape[0]
        if self.tree_init:
            if verbose:
                print(ts(), "Building RP forest with", str(n_trees), "trees")
            eff_leaf_size = leaf_size
            if eff_leaf_size is None:
                eff_leaf_size = max(60, min(256, 5 * int(n_neighbors)))  # rp_trees.py:2845-2846
            tree_states = current_random_state.randint(INT32_MIN, INT32_MAX, size=(n_trees, 3)).astype(np.int64)
            eff_trees = n_trees
        else:
            eff_leaf_size = max(60, min(256, 5 * int(n_neighbors))) if leaf_size is None else leaf_size
            tree_states = np.zeros((1, 3), np.int64)
            eff_trees = 0

        if self.max_candidates is None:
            effective_max_candidates = min(60, self.n_neighbors)  # pynndescent_.py:1135-1138
        else:
            effective_max_candidates = self.max_candidates

        if init_graph is not None:
            init_graph = np.asarray(init_graph)
            if init_graph.shape[0] != n:
                raise ValueError("Init graph size does not match dataset size!")  # pynndescent_.py:1229
            if init_dist is not None and init_graph.shape != np.asarray(init_dist).shape:
                raise ValueError("The shapes of init graph and init distances do not match!")  # pynndescent_.py:1236

        if self.n_devices > 1:  # (round 5: a build that starts from init_graph is sharded too -- nnd_build_multi_from_graph)
            self._build_multi(data, metric, n_trees, eff_trees, eff_leaf_size, effective_max_candidates, n_iters, delta,
                              max_rptree_depth, tree_states, verbose, init_graph, init_dist)
        else:
            self._build_single(data, metric, n_trees, eff_trees, eff_leaf_size, effective_max_candidates, n_iters, delta,
                               max_rptree_depth, tree_states, init_graph, init_dist, verbose, device)

        # pynndescent_.py:1262-1267 `np.any(indices < 0)`: rows are ascending with the unfilled entries (-1, +inf) at the
        # tail, so the last column tells (1 M strided reads instead of a 15 M-element temporary)
        if self._neighbor_graph[0][:, -1].min() < 0:
            warn(
                "Failed to correctly find n_neighbors for some samples."
                " Results may be less than ideal. Try re-running with"
                " different parameters."
            )

    def _build_multi(self, data, metric, n_trees, eff_trees, eff_leaf_size, effective_max_candidates, n_iters, delta,
                     max_rptree_depth, tree_states, verbose, init_graph=None, init_dist=None):
        """Row-sharded build over several GPUs, one call into the library (include/pynnd_amd.h nnd_build_multi)."""
        from sklearn.utils import assert_all_finite

        from . import sharded

        assert_all_finite(data)  # check_array's scan (pynndescent_.py:1054): the one-call multi-GPU build has no flag to read
        if verbose:
            print(ts(), "NN descent for", str(n_iters), "iterations on", self.n_devices, "GPUs")
        idx, dst, st, info = sharded.build_multi(
            data, self.n_devices, self.devices, metric, self.n_neighbors, eff_trees, eff_leaf_size, effective_max_candidates,
            n_iters, delta, max_rptree_depth=max_rptree_depth, rng_state=self.rng_state, tree_state=tree_states[0],
            init_graph=init_graph, init_dist=init_dist)
        self._rp_forest = _DeviceForestSentinel(n_trees, st["n_leaves"], eff_leaf_size) if self.tree_init else None
        self._neighbor_graph = (idx, dst)
        self._build_stats = st
        self._shard_info = info
        if verbose:
            for it, c in enumerate(info["c"]):
                print("\t", it + 1, " / ", n_iters, " c =", c)

    def _build_single(self, data, metric, n_trees, eff_trees, eff_leaf_size, effective_max_candidates, n_iters, delta,
                      max_rptree_depth, tree_states, init_graph, init_dist, verbose, device):
        n, n_neighbors = data.shape[0], self.n_neighbors
        builder = _capi.Builder(
            n, data.shape[1], _METRIC_CODES[metric], n_neighbors, eff_trees, eff_leaf_size, max_rptree_depth,
            effective_max_candidates, n_iters, delta, self.rng_state, tree_states[0], device=device,
        )
        try:
            builder.set_data_host(data)
            _raise_if_nonfinite(builder, data)
            if self.tree_init:
                builder.make_forest()
                st = builder.stats()
                self._rp_forest = _DeviceForestSentinel(n_trees, st["n_leaves"], eff_leaf_size)
            else:
                self._rp_forest = None
            if verbose:
                print(ts(), "NN descent for", str(n_iters), "iterations")
            if init_graph is None:
                if self.tree_init:
                    builder.init_from_leaves()
                builder.init_random()
            else:
                builder.init_from_graph(init_graph, init_dist)
            if verbose:
                # nn_descent_internal (pynndescent_.py:296-320), driven from here so that the output matches the reference's
                for it in range(n_iters):
                    print("\t", it + 1, " / ", n_iters)
                    c = builder.descent_iter()
                    if c <= delta * n_neighbors * n:
                        print("\tStopping threshold met -- exiting after", it + 1, "iterations")
                        break
            else:
                builder.descent()  # the same loop inside the library (nnd_descent): no host round trip per iteration
            self._neighbor_graph = builder.finalize()
            self._build_stats = builder.stats()
        finally:
            builder.close()

    @property
    def neighbor_graph(self):
        """pynndescent_.py:2145-2158: copies, with the distance correction applied."""
        if self.compressed and not hasattr(self, "_neighbor_graph"):
            warn("Compressed indexes do not have neighbor graph information.")
            return None
        return (_capi.host_copy(self._neighbor_graph[0]), self._distance_correction(self._neighbor_graph[1]))

    def build_search_graph(self):
        """The pruning pass of ``_init_search_graph`` (pynndescent_.py:1451-1611: diversify, reverse diversify,
        degree prune) on the GPU alone; returns the CSR uint8 graph in the ORIGINAL vertex numbering (``prepare()``
        runs the same pass and then re-indexes everything by the hub search tree's leaf order)."""
        from .search_graph import build_search_graph

        if hasattr(self, "_vertex_order"):
            raise RuntimeError("the index is prepared already: its search graph is index._search_graph (in leaf order)")
        self._pruned_graph = build_search_graph(
            self._raw_data, self._neighbor_graph[0], self._neighbor_graph[1], self.metric, self.n_neighbors,
            self.prune_degree_multiplier, self.diversify_prob, self.diversify_method,
            degree_prune_aggressiveness=self.degree_prune_aggressiveness,
            seed=int(self.rng_state[0]) & 0xFFFFFFFF, device=self.device)
        return self._pruned_graph

    # attributes of the reference class that prepare() / query() / update() read (pynndescent_.py:1014-1113)
    _HANDOVER = (
        "n_trees", "angular_trees", "n_trees_after_update", "n_neighbors", "metric", "metric_kwds", "bit_metric",
        "leaf_size", "prune_degree_multiplier", "diversify_prob", "diversify_method", "degree_prune_aggressiveness",
        "n_search_trees", "search_tree_leaf_size", "max_search_tree_depth", "max_rptree_depth", "max_candidates",
        "quantization", "low_memory", "n_iters", "delta", "dim", "n_jobs", "compressed", "parallel_batch_queries",
        "verbose", "_input_dtype", "_raw_data", "tree_init", "_dist_args", "random_state", "_angular_trees",
        "_bit_trees", "_is_sparse", "rng_state", "search_rng_state", "_rp_forest",
    )

    def to_reference(self):
        """Hand the GPU-built index to ``pynndescent.NNDescent`` (must be importable) WITHOUT rebuilding: the returned
        object carries this index's data, graph and parameters, so the reference's own ``prepare()`` (hub search tree,
        pruned + reordered search graph) and ``query()`` run on it unchanged.  The reference only null-checks
        ``_rp_forest`` before building its hub tree from the graph (pynndescent_.py:1353-1437)."""
        import pynndescent

        ref = object.__new__(pynndescent.NNDescent)
        for name in self._HANDOVER:
            if hasattr(self, name):
                setattr(ref, name, getattr(self, name))
        if hasattr(self, "_neighbor_graph"):
            ref._neighbor_graph = (self._neighbor_graph[0].copy(), self._neighbor_graph[1].copy())
        ref._distance_correction = None
        ref._set_distance_func()  # pynndescent_.py:1271-1298: numba distance, correction, proxy flags
        if hasattr(self, "_search_graph"):  # prepared on the GPU: the reference's prepare() has nothing left to build
            from pynndescent.rp_trees import FlatTree as RefFlatTree

            ref._search_graph = self._search_graph.copy()
            ref._search_forest = [RefFlatTree(*t) for t in self._search_forest]
            ref._vertex_order = np.asarray(self._vertex_order).copy()
            ref._min_distance = self._min_distance
            ref._visited = np.zeros_like(self._visited)
            if hasattr(ref, "_rp_forest"):
                del ref._rp_forest
        return ref

    @classmethod
    def from_graph(cls, data, indices, distances, metric="euclidean", random_state=None, **kwargs):
        """An index object around an existing k-NN graph (``distances`` in the alternative space, rows ascending), no
        build: the starting point for ``update()`` / ``build_search_graph()`` / ``to_reference()``."""
        self = object.__new__(cls)
        data = check_array(data, dtype=np.float32, order="C")
        n = data.shape[0]
        n_trees = kwargs.pop("n_trees", None)
        n_iters = kwargs.pop("n_iters", None)
        if n_trees is None:
            n_trees = max(3, min(12, int(round(2.0 * np.log10(n)))))
        if n_iters is None:
            n_iters = max(5, int(round(np.log2(n))))
        defaults = dict(
            angular_trees=False, metric_kwds=None, bit_metric=False, leaf_size=None, prune_degree_multiplier=1.5,
            diversify_prob=1.0, diversify_method="standard", degree_prune_aggressiveness=1.0, n_search_trees=1,
            search_tree_leaf_size=None, max_search_tree_depth=None, max_rptree_depth=200, max_candidates=None,
            quantization=None, low_memory=True, delta=0.001, n_jobs=None, compressed=False,
            parallel_batch_queries=False, verbose=False, device=0,
        )
        if "pruning_degree_multiplier" in kwargs:  # the constructor's name for prune_degree_multiplier
            kwargs["prune_degree_multiplier"] = kwargs.pop("pruning_degree_multiplier")
        unknown = set(kwargs) - set(defaults)
        if unknown:
            raise TypeError("unexpected arguments: %s" % sorted(unknown))
        defaults.update(kwargs)
        for name, value in defaults.items():
            setattr(self, name, value)
        if metric not in _METRIC_CODES:
            raise ValueError("Metric is neither callable, " + "nor a recognised string")
        indices = np.ascontiguousarray(indices, np.int32)
        distances = np.ascontiguousarray(distances, np.float32)
        if indices.shape != distances.shape or indices.shape[0] != n:
            raise ValueError("Init graph size does not match dataset size!")
        _check_supported_sizes(indices.shape[1], self.max_candidates, None)
        self.metric, self.n_neighbors = metric, indices.shape[1]
        self.n_trees, self.n_iters = n_trees, n_iters
        self.n_trees_after_update = max(2, int(np.round(n_trees / 3)))
        self.dim = data.shape[1]
        self._input_dtype, self._raw_data = np.float32, data
        self.tree_init = True
        self._dist_args = tuple((self.metric_kwds or {}).values())
        self.random_state = random_state
        rs = check_random_state(random_state)
        self._distance_correction = _DISTANCE_CORRECTIONS[metric]
        self._distance_func = None
        self._angular_trees = metric in _ANGULAR_METRICS
        self._bit_trees = self._is_sparse = False
        self.rng_state = rs.randint(INT32_MIN, INT32_MAX, 3).astype(np.int64)
        self.search_rng_state = rs.randint(INT32_MIN, INT32_MAX, 3).astype(np.int64)
        for _ in range(10):
            tau_rand_int(self.search_rng_state)
        self._rp_forest = _DeviceForestSentinel(n_trees, 0, 0)
        self._neighbor_graph = (indices, distances)
        return self

    # ------------------------------------------------------------------------------------------------ prepare / query
    def _init_search_graph(self):
        """``NNDescent._init_search_graph`` (pynndescent_.py:1333-1662) on the GPU: the hub search tree built from the
        data and the finished graph (csrc/hubtree.hip), the pruning pass (csrc/prune.hip), then data, graph and tree
        re-indexed by the tree's leaf order (pynndescent_.py:1629-1651)."""
        from .search_graph import build_search_graph
        from .search_tree import make_hub_tree, reorder_by_tree

        search_leaf_size = (self.search_tree_leaf_size if self.search_tree_leaf_size is not None
                            else (self.leaf_size if self.leaf_size is not None else 30))  # pynndescent_.py:1341-1345
        search_tree_depth = (self.max_search_tree_depth if self.max_search_tree_depth is not None
                             else self.max_rptree_depth)                                   # pynndescent_.py:1346-1350
        if not hasattr(self, "_search_forest"):
            if getattr(self, "_rp_forest", None) is None and not self.tree_init:
                self._search_forest = []  # pynndescent_.py:1377-1378: no tree, queries start from random vertices
            else:
                if self.verbose:
                    print(ts(), "Building hub-based search tree")
                self._search_forest = [make_hub_tree(self._raw_data, self._neighbor_graph[0], self.metric, search_leaf_size,
                                                     search_tree_depth, device=self.device,
                                                     seed=int(self.rng_state[0]) & 0x7FFFFFFF)]
                self._rp_forest = None  # the reference deletes it here (pynndescent_.py:1444)
        if self.verbose:
            print(ts(), "Diversifying and pruning the search graph")
        graph, stages = build_search_graph(
            self._raw_data, self._neighbor_graph[0], self._neighbor_graph[1], self.metric, self.n_neighbors,
            self.prune_degree_multiplier, self.diversify_prob, self.diversify_method,
            degree_prune_aggressiveness=self.degree_prune_aggressiveness, seed=int(self.rng_state[0]) & 0xFFFFFFFF,
            device=self.device, return_stages=True)
        self._min_distance = np.float32(stages["min_distance"])                     # pynndescent_.py:1539
        self._visited = np.zeros((self._raw_data.shape[0] // 8) + 1, dtype=np.uint8, order="C")  # pynndescent_.py:1624-1626
        if self.verbose:
            print(ts(), "Resorting data and graph based on tree order")
        if self._search_forest:
            graph, data, self._vertex_order, tree = reorder_by_tree(graph, self._raw_data, self._search_forest[0])
            self._raw_data = data
            self._search_forest = [tree] + list(self._search_forest[1: self.n_search_trees])
        else:
            self._vertex_order = np.arange(self._raw_data.shape[0])
        self._search_graph = graph
        self._searcher = None
        if self.compressed:  # pynndescent_.py:1653-1658
            if hasattr(self, "_rp_forest"):
                del self._rp_forest
            del self._neighbor_graph

    def prepare(self):
        """``NNDescent.prepare`` (pynndescent_.py:2174-2273): build everything a query needs."""
        if self.quantization is not None:
            raise NotImplementedError("quantized search (quantization=%r) is out of scope for pynndescent_amd; use "
                                      "index.to_reference()" % (self.quantization,))
        if not hasattr(self, "_search_graph"):
            self._init_search_graph()
        if getattr(self, "_searcher", None) is None:
            tree = self._search_forest[0] if self._search_forest else None
            self._searcher = _capi.Searcher(self._raw_data, self._search_graph, tree, _METRIC_CODES[self.metric],
                                            self._min_distance, self.n_neighbors, self.search_rng_state, device=self.device)

    def query(self, query_data, k=10, epsilon=0.1, proxy_beam_size=4):
        """``NNDescent.query`` (pynndescent_.py:2275-2379) on the GPU: one wave per query (csrc/query.hip).
        Returns (indices (n_queries, k) in the ORIGINAL numbering, true distances (n_queries, k))."""
        if k > 256:
            raise NotImplementedError("pynndescent_amd answers queries with k <= 256; use index.to_reference() for k = %d" % k)
        if not hasattr(self, "_search_graph") or getattr(self, "_searcher", None) is None:
            self.prepare()
        query_data = np.asarray(query_data).astype(np.float32, order="C")  # pynndescent_.py:2316
        if query_data.ndim != 2 or query_data.shape[1] != self._raw_data.shape[1]:
            raise ValueError("query_data must have shape (n_queries, %d)" % self._raw_data.shape[1])
        indices, dists = self._searcher.query(query_data, k, epsilon)
        found = indices >= 0
        indices = np.where(found, self._vertex_order[np.where(found, indices, 0)], -1).astype(np.int32)  # pynndescent_.py:2373
        if self._distance_correction is not None:  # pynndescent_.py:2375-2376
            dists = self._distance_correction(dists)
        return indices, dists

    # ------------------------------------------------------------------------------------------------ pickling
    def __getstate__(self):
        """pynndescent_.py:1306-1320: a pickled index is a PREPARED index; device handles and the build forest stay behind."""
        if not hasattr(self, "_search_graph"):
            self._init_search_graph()
        state = self.__dict__.copy()
        state.pop("_rp_forest", None)
        state.pop("_searcher", None)
        state["_search_forest"] = tuple(tuple(t) for t in self._search_forest)  # rp_trees.py:3060-3069 denumbaify_tree
        return state

    def __setstate__(self, d):
        """pynndescent_.py:1322-1331: rebuild what cannot be pickled (here: the device copy, lazily on the first query)."""
        from .search_tree import FlatTree

        self.__dict__ = d
        self._distance_correction = _DISTANCE_CORRECTIONS[self.metric]
        self._search_forest = [FlatTree(*t) for t in d["_search_forest"]]  # rp_trees.py:3072-3081 renumbaify_tree
        self._searcher = None

    def update(self, xs_fresh=None, xs_updated=None, updated_indices=None):
        """``pynndescent.NNDescent.update`` (pynndescent_.py:2381-2553) on the GPU: fresh rows are appended, updated
        rows replaced (their graph rows and every edge pointing at them are dropped), then the graph is rebuilt from
        a warm start -- the old graph inserted as "old" edges (init_from_neighbor_graph, flag 0), a forest of
        ``n_trees_after_update`` trees seeding "new" edges, no random fill -- and NN-descent runs to the stop rule."""
        current_random_state = check_random_state(self.random_state)
        # drawn and handed to make_forest by the reference (pynndescent_.py:2408-2411); kept for the stream position
        current_random_state.randint(INT32_MIN, INT32_MAX, 3)
        if xs_updated is not None:
            xs_updated = check_array(xs_updated, dtype=self._input_dtype, order="C")
            if updated_indices is None:
                raise ValueError("If xs_updated are provided, updated_indices must also be provided!")
            try:
                updated_indices = list(map(int, updated_indices))
            except (TypeError, ValueError):
                raise ValueError("Could not convert updated indices to list of int(s).")
            n1, n2 = len(updated_indices), xs_updated.shape[0]
            if n1 != n2:
                raise ValueError(
                    f"Number of updated indices ({n1}) must match " f"number of rows of xs_updated ({n2})."
                )
        else:
            if updated_indices is not None:
                warn("xs_updated not provided, while update_indices provided. " "They will be ignored.")
            updated_indices = None
        if updated_indices is None:
            xs_updated = np.zeros((0, self._raw_data.shape[1]), self._input_dtype)
            updated_indices = []
        if xs_fresh is None:
            xs_fresh = np.zeros((0, self._raw_data.shape[1]), dtype=self._input_dtype)
        else:
            xs_fresh = check_array(xs_fresh, dtype=self._input_dtype, order="C")

        # data and graph invalidation (pynndescent_.py:2461-2493), vectorised; a prepared index keeps its rows in the
        # search tree's leaf order: back to the original order first (pynndescent_.py:2462-2465, 2476)
        raw = np.array(self._raw_data, copy=True)
        if hasattr(self, "_vertex_order"):
            raw = raw[np.argsort(self._vertex_order), :]
        for x_updated, i_fresh in zip(xs_updated, updated_indices):
            raw[i_fresh] = x_updated
        n_old = raw.shape[0]
        raw = np.ascontiguousarray(np.vstack([raw, xs_fresh]))
        ns, ds = (np.array(a, copy=True) for a in self._neighbor_graph)
        if updated_indices:
            hit = np.zeros(n_old, bool)
            hit[updated_indices] = True
            ns[hit] = -1
            ds[hit] = np.inf
            stale = (ns >= 0) & hit[np.clip(ns, 0, None)]
            ns[stale] = -1
            ds[stale] = np.inf
        n = raw.shape[0]
        pad_i = np.full((n, ns.shape[1]), -1, np.int32)
        pad_d = np.full((n, ns.shape[1]), np.inf, np.float32)
        pad_i[:n_old] = ns
        pad_d[:n_old] = ds

        self.n_trees = self.n_trees_after_update  # pynndescent_.py:2498
        eff_leaf_size = self.leaf_size
        if eff_leaf_size is None:
            eff_leaf_size = max(60, min(256, 5 * int(self.n_neighbors)))  # rp_trees.py:2845-2846
        tree_states = current_random_state.randint(INT32_MIN, INT32_MAX, size=(self.n_trees, 3)).astype(np.int64)
        if self.max_candidates is None:
            effective_max_candidates = min(60, self.n_neighbors)
        else:
            effective_max_candidates = self.max_candidates
        if getattr(self, "n_devices", 1) > 1:  # round 5: the rebuild is sharded like the build was (nnd_build_multi_update)
            from sklearn.utils import assert_all_finite

            from . import sharded

            assert_all_finite(raw)
            idx, dst, st, info = sharded.build_multi(
                raw, self.n_devices, self.devices, self.metric, self.n_neighbors, self.n_trees, eff_leaf_size, effective_max_candidates,
                self.n_iters, self.delta, max_rptree_depth=self.max_rptree_depth, rng_state=self.rng_state, tree_state=tree_states[0],
                old_graph=(pad_i, pad_d))
            self._rp_forest = _DeviceForestSentinel(self.n_trees, st["n_leaves"], eff_leaf_size)
            self._neighbor_graph = (idx, dst)
            self._build_stats = st
            self._shard_info = info
            self._raw_data = raw
            if hasattr(self, "_search_graph"):
                for name in ("_search_graph", "_search_forest", "_vertex_order", "_searcher"):
                    if hasattr(self, name):
                        delattr(self, name)
                self.prepare()
            return
        builder = _capi.Builder(
            n, raw.shape[1], _METRIC_CODES[self.metric], self.n_neighbors, self.n_trees, eff_leaf_size,
            self.max_rptree_depth, effective_max_candidates, self.n_iters, self.delta, self.rng_state, tree_states[0],
            device=self.device,
        )
        try:
            builder.set_data_host(raw)
            builder.make_forest()
            self._rp_forest = _DeviceForestSentinel(self.n_trees, builder.stats()["n_leaves"], eff_leaf_size)
            builder.reset_graph()
            builder.init_from_neighbor_graph(pad_i, pad_d)  # pynndescent_.py:2512-2516
            builder.init_from_leaves()                      # pynndescent_.py:2517
            for it in range(self.n_iters):                  # nn_descent with init_graph: no random fill (P_:346-350)
                if self.verbose:
                    print("\t", it + 1, " / ", self.n_iters)
                c = builder.descent_iter()
                if c <= self.delta * self.n_neighbors * n:
                    if self.verbose:
                        print("\tStopping threshold met -- exiting after", it + 1, "iterations")
                    break
            self._neighbor_graph = builder.finalize()
            self._build_stats = builder.stats()
        finally:
            builder.close()
        self._raw_data = raw
        if hasattr(self, "_search_graph"):  # pynndescent_.py:2538-2553: the derived structures are rebuilt
            for name in ("_search_graph", "_search_forest", "_vertex_order", "_searcher"):
                if hasattr(self, name):
                    delattr(self, name)
            self.prepare()


EMPTY_GRAPH = (np.array([[-1]], dtype=np.int32), np.array([[np.inf]], dtype=np.float32),
               np.array([[0]], dtype=np.uint8))  # pynndescent_.py:64-68

# distance arguments nn_descent understands: name (or __name__ of the reference's function) -> (kernel metric, correction)
_ND_DISTS = {"squared_euclidean": (_capi.NND_METRIC_SQEUCLIDEAN, None), "sqeuclidean": (_capi.NND_METRIC_SQEUCLIDEAN, None),
             "euclidean": (_capi.NND_METRIC_SQEUCLIDEAN, np.sqrt), "l2": (_capi.NND_METRIC_SQEUCLIDEAN, np.sqrt),
             "alternative_cosine": (_capi.NND_METRIC_ALT_COSINE, None),
             "cosine": (_capi.NND_METRIC_ALT_COSINE, correct_alternative_cosine)}


def nn_descent(data, n_neighbors, rng_state, max_candidates=50, dist="squared_euclidean", n_iters=10, delta=0.001,
               init_graph=EMPTY_GRAPH, rp_tree_init=True, leaf_array=None, low_memory=True, verbose=False, device=0):
    """The reference's ``nn_descent`` (pynndescent_.py:323-366) with the same arguments, on the GPU: the seam a caller
    uses who keeps the reference's ``make_forest`` / ``rptree_leaf_array`` and hands the leaves in.

    data float32 (n, d); rng_state int64[3]; ``dist``: "squared_euclidean" / "alternative_cosine" (what NNDescent passes,
    pynndescent_.py:1247-1260), "euclidean" / "cosine" (true distances: the same kernels, corrected on return), or the
    reference function of one of those names; ``init_graph``: EMPTY_GRAPH, or the heap triple ``(indices (n, k),
    distances (n, k), flags (n, k))`` the reference accepts (entries with index -1 are empty); ``leaf_array`` int32
    (n_leaves, max_leaf_size), -1 padded, used when ``rp_tree_init``.  ``low_memory`` is accepted and unused, as in the
    reference (pynndescent_.py:335).  Returns ``(indices int32 (n, k), distances float32 (n, k))``, rows ascending."""
    name = dist if isinstance(dist, str) else getattr(dist, "__name__", None)
    if name not in _ND_DISTS:
        raise NotImplementedError("pynndescent_amd.nn_descent: dist must be one of %s (got %r)" % (sorted(_ND_DISTS), dist))
    code, correction = _ND_DISTS[name]
    data = np.ascontiguousarray(data, dtype=np.float32)
    n = data.shape[0]
    _check_supported_sizes(n_neighbors, max_candidates, None)
    empty = init_graph[0].shape[0] == 1  # EMPTY_GRAPH
    if not empty and not (init_graph[0].shape[0] == n and init_graph[0].shape[1] == n_neighbors):
        raise ValueError("Invalid initial graph specified!")  # pynndescent_.py:352
    builder = _capi.Builder(n, data.shape[1], code, n_neighbors, 0, max(60, min(256, 5 * int(n_neighbors))), 200,
                            max_candidates, n_iters, delta, np.asarray(rng_state, np.int64), np.zeros(3, np.int64),
                            device=device)
    try:
        builder.set_data_host(data)
        if empty:
            if rp_tree_init:
                if leaf_array is None:
                    raise ValueError("rp_tree_init=True needs a leaf_array (rptree_leaf_array of a forest)")
                builder.init_from_leaf_array(leaf_array)
            builder.init_random()
        else:  # a heap handed over: its entries, with their distances (flags restart as "new": they were never sampled here)
            builder.init_from_graph(np.asarray(init_graph[0], np.int32), np.asarray(init_graph[1], np.float32))
        for it in range(n_iters):  # nn_descent_internal (pynndescent_.py:296-320)
            if verbose:
                print("\t", it + 1, " / ", n_iters)
            c = builder.descent_iter()
            if c <= delta * n_neighbors * n:
                if verbose:
                    print("\tStopping threshold met -- exiting after", it + 1, "iterations")
                break
        idx, dst = builder.finalize()
    finally:
        builder.close()
    if correction is not None:
        dst = correction(dst).astype(np.float32)
    return idx, dst


def _check_array_no_scan(data):
    try:
        return check_array(data, dtype=np.float32, order="C", ensure_all_finite=False)
    except TypeError:  # scikit-learn < 1.6
        return check_array(data, dtype=np.float32, order="C", force_all_finite=False)


def _raise_if_nonfinite(builder, data):
    """The ValueError check_array raises in the reference for NaN / inf input (pynndescent_.py:1054), from the device flag."""
    if builder.data_nonfinite():
        from sklearn.utils import assert_all_finite

        assert_all_finite(data)  # raises "Input contains NaN." / "... infinity or a value too large ..."
        raise ValueError("Input contains NaN or infinity.")  # (unreachable unless the two scans disagree)


def _check_supported_sizes(n_neighbors, max_candidates, init_graph):
    """The GPU k-lists hold at most 256 entries (four per lane of a wave; rows above 64 take the LDS-merge kernels) and the
    candidate lists at most 128 (above 64: five passes of the 64-slot join over blocks of the lists); the reference has no such bounds (pynndescent_.py:976-982), so the limits are reported up
    front and by name."""
    if int(n_neighbors) > 256 or (max_candidates is not None and int(max_candidates) > 128):
        raise NotImplementedError(
            "pynndescent_amd supports n_neighbors <= 256 and max_candidates <= 128 (got n_neighbors=%s, max_candidates=%s); "
            "use pynndescent.NNDescent (or pynndescent_amd.make_index) for wider graphs" % (n_neighbors, max_candidates))
    if init_graph is not None and np.ndim(init_graph) == 2 and np.shape(init_graph)[1] > 256:
        raise NotImplementedError("pynndescent_amd supports init_graph with at most 256 columns (got %d); use "
                                  "pynndescent.NNDescent" % np.shape(init_graph)[1])


def make_index(data, *args, **kwargs):
    """``NNDescent(data, ...)`` on the GPU when the input is in scope (dense data, euclidean / l2 / sqeuclidean / cosine, k <= 256);
    otherwise -- and only then -- the reference ``pynndescent.NNDescent`` on the CPU when that package is importable
    (SURVEY.md section 8b), with a warning.  A missing HIP library or GPU is never papered over: that still raises."""
    device = kwargs.pop("device", 0)
    try:
        return NNDescent(data, *args, device=device, **kwargs)
    except NotImplementedError as exc:
        try:
            import pynndescent
        except ImportError:
            raise exc
        warn("pynndescent_amd: %s -- building with pynndescent.NNDescent on the CPU instead" % exc)
        return pynndescent.NNDescent(data, *args, **kwargs)


# string metrics the reference recognises (distances.py:2103-2168 named_distances keys) -- used only to
# decide between NotImplementedError (valid in the reference, not accelerated) and the reference's ValueError.
_KNOWN_REFERENCE_METRICS = frozenset(
    """euclidean l2 sqeuclidean manhattan taxicab l1 chebyshev linfinity linfty linf minkowski seuclidean
    standardised_euclidean wminkowski weighted_minkowski mahalanobis canberra cosine dot inner_product correlation
    haversine braycurtis spearmanr tsss true_angular hellinger kantorovich wasserstein wasserstein_1d
    wasserstein-1d kantorovich-1d kantorovich_1d circular_kantorovich circular_wasserstein sinkhorn jensen-shannon
    jensen_shannon symmetric-kl symmetric_kl symmetric_kullback_liebler hamming jaccard dice matching kulsinski
    rogerstanimoto russellrao sokalsneath sokalmichener yule bit_hamming bit_jaccard""".split()
)

"""Search-graph pruning pass on the GPU (BASELINE config 5: "+ graph diversification/prune pass").

Mirrors the pruning part of ``NNDescent._init_search_graph`` (reference pynndescent_.py:1451-1611):
forward ``diversify`` (369-403) or ``diversify_degree_aware`` (433-546) -> COO -> CSR -> "reverse"
``diversify_csr`` (549-588) or ``diversify_csr_degree_aware`` (625-726) -> union by element-wise maximum -> drop
the diagonal -> ``degree_prune`` to ``round(pruning_degree_multiplier * n_neighbors)`` (728-760) -> binarise.
The numba kernels run as HIP kernels (csrc/prune.hip); the conversions between them are the same scipy calls the
reference makes.  Two things the reference does BY ACCIDENT are reproduced, because parity is with what it computes:

* scipy's ``transpose()`` of a CSR matrix is a CSC matrix over the SAME ``indptr`` / ``indices`` / ``data`` arrays,
  so the "reverse" pass walks the forward rows again;
* and because the arrays are shared, zeroing weights in the "reverse" matrix and calling its ``eliminate_zeros()``
  (which compacts in place) prunes the FORWARD matrix as well: the union is ``max(F', F'^T)`` of the doubly pruned
  ``F'``, not ``max(F, F'^T)`` (pynndescent_.py:1541-1599; verified with scipy 1.15.3).  With ``diversify_prob = 1``
  the second pass finds nothing new on rows the first pass already diversified; with ``diversify_prob < 1`` it
  re-tests the edges the first pass's coins spared.

The hub search tree and the reordering by its leaf order (1629-1651) are in ``pynndescent_amd/search_tree.py``.
"""
import numpy as np
import scipy.sparse as sp

from . import _capi

FLOAT32_EPS = np.finfo(np.float32).eps  # pynndescent_.py:65
DEVICE_PASS_MAX_EDGES = 0x7FFFFFF0      # csrc/searchgraph.hip: 2 n k keyed edges, int32 positions


def compute_degrees(indices):
    """pynndescent_.py:406-418: undirected degree of every vertex of an (n,k) neighbour array."""
    idx = np.asarray(indices)
    n = idx.shape[0]
    valid = idx >= 0
    deg = valid.sum(1).astype(np.int64)
    deg += np.bincount(idx[valid].ravel(), minlength=n)[:n]
    return deg.astype(np.int32)


def compute_degrees_csr(indptr, indices):
    """pynndescent_.py:591-622."""
    n = indptr.shape[0] - 1
    deg = np.diff(indptr).astype(np.int64)
    ind = np.asarray(indices[: indptr[-1]])
    deg += np.bincount(ind[(ind >= 0) & (ind < n)], minlength=n)[:n]
    return deg.astype(np.int32)


def search_graph_on(builder, indices, distances, n_neighbors=None, pruning_degree_multiplier=1.5, diversify_prob=1.0,
                    diversify_method="standard", degree_prune_aggressiveness=1.0, seed=0, on_device=False, return_stages=False):
    """The pass on a handle that already holds the point set (``_capi.Builder`` after ``set_data_*`` -- e.g. the one that has
    just built the graph: no second upload of the rows).  ``indices`` / ``distances``: numpy (n, k) arrays, or device addresses
    with ``on_device=True`` (the graph as ``build_device`` left it: it never visits the host).  Everything between the
    kernels -- COO -> CSR, the transpose, the maximum, the diagonal, the zeros, the binarisation -- runs on the device
    (csrc/searchgraph.hip); ONE copy of ``indptr`` / ``indices`` comes back."""
    if diversify_method not in ("standard", "degree_aware"):
        raise ValueError("diversify_method must be 'standard' or 'degree_aware'")
    n, k = builder.n, builder.k
    if k > 256:
        raise NotImplementedError("the pruning pass (prepare / query) handles graphs of at most 256 neighbours per row (got %d)" % k)
    n_neighbors = k if n_neighbors is None else n_neighbors
    res = builder.search_graph(indices, distances, n_neighbors, pruning_degree_multiplier, diversify_prob,
                               diversify_method == "degree_aware", degree_prune_aggressiveness, seed, on_device=on_device,
                               want_forward=return_stages)
    indptr, ind, st = res[:3]
    graph = sp.csr_array((np.ones(ind.shape[0], np.uint8), ind, indptr), shape=(n, n))
    graph.has_sorted_indices = True
    if return_stages:
        return graph, {"forward_rows": res[3], "forward_dist": res[4], "nnz_pre_diversify": None, "forward_nnz": int(st["forward_nnz"]),
                       "reverse_nnz": int(st["reverse_nnz"]), "union_nnz": int(st["union_nnz"]), "final_nnz": int(st["final_nnz"]),
                       "min_distance": float(st["min_distance"]), "ms_device": float(st["ms_device"])}
    return graph


def build_search_graph(data, indices, distances, metric="euclidean", n_neighbors=None, pruning_degree_multiplier=1.5,
                       diversify_prob=1.0, diversify_method="standard", degree_prune_aggressiveness=1.0, seed=0,
                       device=0, return_stages=False, host_glue=False):
    """(data, indices int32 (n,k), alt-space distances float32 (n,k)) -> scipy CSR uint8 search graph (unordered): an auxiliary
    handle for the point set, then ``search_graph_on``.  ``host_glue=True``: the rounds 2-4 form (the three kernels of
    csrc/prune.hip with the reference's scipy calls between them), kept as the comparison for the device pass."""
    if not host_glue:
        if diversify_method not in ("standard", "degree_aware"):
            raise ValueError("diversify_method must be 'standard' or 'degree_aware'")
        if np.shape(indices)[1] > 256:
            raise NotImplementedError("the pruning pass (prepare / query) handles graphs of at most 256 neighbours per row (got %d)"
                                      % np.shape(indices)[1])
        x = np.ascontiguousarray(data, dtype=np.float32)
        n, d = x.shape
        k = np.shape(indices)[1]
        code = _capi.METRIC_CODES[metric]
        # The device pass keeps 2 n k keyed edges with int32 positions and ~104 bytes of workspace per edge; a graph beyond that
        # (or a device too full for the workspace) goes through the kernels with the reference's scipy calls between them --
        # the rounds 2-4 form, same edges (tests/test_gpu_build.py), n k < 2^31 -- instead of failing prepare().
        fits = 2 * n * k < DEVICE_PASS_MAX_EDGES
        b = _capi.Builder(n, d, code, k, 0, 60, 200, min(60, k), 1, 0.001, [1, 2, 3], [4, 5, 6], device=device, flags=_capi.NND_FLAG_NO_GRAPH) if fits else None
        try:
            if fits:
                b.set_data_host(x)
                try:
                    out = search_graph_on(b, indices, distances, n_neighbors, pruning_degree_multiplier, diversify_prob, diversify_method,
                                          degree_prune_aggressiveness, seed, return_stages=return_stages)
                    if return_stages:
                        out[1]["nnz_pre_diversify"] = int((np.asarray(indices) >= 0).sum())
                    return out
                except _capi.NNDError as e:
                    if "memory" not in str(e).lower():
                        raise
                    import warnings

                    warnings.warn("the search-graph pass does not fit the device (%s): running it with host-side sparse-matrix steps" % e)
        finally:
            if b is not None:
                b.close()
    return _build_search_graph_host_glue(data, indices, distances, metric, n_neighbors, pruning_degree_multiplier, diversify_prob,
                                         diversify_method, degree_prune_aggressiveness, seed, device, return_stages)


def _build_search_graph_host_glue(data, indices, distances, metric="euclidean", n_neighbors=None, pruning_degree_multiplier=1.5,
                                  diversify_prob=1.0, diversify_method="standard", degree_prune_aggressiveness=1.0, seed=0,
                                  device=0, return_stages=False):
    """(indices int32 (n,k), alt-space distances float32 (n,k)) -> scipy CSR uint8 search graph (unordered).

    ``indices``/``distances`` are ``NNDescent._neighbor_graph`` (rows ascending, squared-L2 / log2-cosine)."""
    if diversify_method not in ("standard", "degree_aware"):
        raise ValueError("diversify_method must be 'standard' or 'degree_aware'")
    if np.shape(indices)[1] > 256:
        raise NotImplementedError("the pruning pass (prepare / query) handles graphs of at most 256 neighbours per row (got %d)"
                                  % np.shape(indices)[1])
    aware = diversify_method == "degree_aware"
    x = np.ascontiguousarray(data, dtype=np.float32)
    n, d = x.shape
    k = indices.shape[1]
    n_neighbors = k if n_neighbors is None else n_neighbors
    code = _capi.METRIC_CODES[metric]
    b = _capi.Builder(n, d, code, k, 0, 60, 200, min(60, k), 1, 0.001, [1, 2, 3], [4, 5, 6], device=device,
                      flags=_capi.NND_FLAG_NO_GRAPH)  # the pass reads rows and norms only: no k-lists / proposal tables
    try:
        b.set_data_host(x)
        nnz_pre = int((np.asarray(indices) >= 0).sum())
        if aware:  # pynndescent_.py:1476-1498: max_degree = int(multiplier * k); diversify_prob lands in `alpha`
            rows, dd = b.diversify(indices, distances, degree=compute_degrees(indices), degree_aware=True,
                                   max_degree=max(1, int(pruning_degree_multiplier * n_neighbors)),
                                   aggressiveness=degree_prune_aggressiveness, alpha=diversify_prob, seed=seed)
        else:      # pynndescent_.py:1499-1518
            rows, dd = b.diversify(indices, distances, prune_probability=diversify_prob, seed=seed)
        dd[dd == 0.0] = FLOAT32_EPS  # preserve distance-0 points (pynndescent_.py:1525)
        # COO -> CSR (1527-1537): entries stay in row order (ascending distance), -1 slots dropped
        keep = rows >= 0
        indptr = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int32)
        f_indices, f_data = rows[keep].astype(np.int32), dd[keep].astype(np.float32)
        forward_nnz = int(f_data.shape[0])
        min_distance = float(f_data.min()) if forward_nnz else 0.0  # self._min_distance (pynndescent_.py:1539)
        # "Reverse graph" (1549-1587): the forward rows again (shared arrays, see the module docstring)
        if aware:  # max_degree = n_neighbors (1567)
            rdata = b.diversify_csr(indptr, f_indices, f_data, degree=compute_degrees_csr(indptr, f_indices),
                                    degree_aware=True, max_degree=n_neighbors, aggressiveness=degree_prune_aggressiveness,
                                    prune_probability=diversify_prob, seed=seed)
        else:
            rdata = b.diversify_csr(indptr, f_indices, f_data, prune_probability=diversify_prob, seed=seed)
        # reverse_graph.eliminate_zeros() (1588) compacts the shared arrays: the forward matrix loses the edges too
        fwd = sp.csr_array((rdata, f_indices, indptr), shape=(n, n))
        fwd.eliminate_zeros()
        rev = fwd.transpose().tocsr()
        rev.sort_indices()
        fwd.sort_indices()
        union = fwd.maximum(rev).tocsr()  # 1599
        union.setdiag(0.0)  # 1602-1603
        union.eliminate_zeros()
        nnz_pre_prune = int(union.nnz)
        max_degree = int(np.round(pruning_degree_multiplier * n_neighbors))  # 1606-1609
        pdata = b.degree_prune(union.indptr, union.data, max_degree)
        union = sp.csr_array((pdata, union.indices, union.indptr), shape=(n, n))
        union.eliminate_zeros()
        graph = (union != 0).astype(np.uint8).tocsr()  # 1611
        graph.sort_indices()
    finally:
        b.close()
    if return_stages:
        return graph, {"forward_rows": rows, "forward_dist": dd, "nnz_pre_diversify": nnz_pre, "forward_nnz": forward_nnz,
                       "reverse_nnz": int(rev.nnz), "union_nnz": nnz_pre_prune, "final_nnz": int(graph.nnz),
                       "min_distance": min_distance}
    return graph

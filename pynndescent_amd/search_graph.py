"""Search-graph pruning pass on the GPU (BASELINE config 5: "+ graph diversification/prune pass").

Mirrors the pruning part of ``NNDescent._init_search_graph`` (reference pynndescent_.py:1451-1611) for the
standard diversify method at ``diversify_prob = 1``: forward ``diversify`` (369-403) -> COO -> CSR ->
"reverse" ``diversify_csr`` (549-588) -> union by element-wise maximum -> drop the diagonal ->
``degree_prune`` to ``round(pruning_degree_multiplier * n_neighbors)`` (728-760) -> binarise.
The three numba kernels run as HIP kernels (csrc/prune.hip); the conversions between them are the same scipy
calls the reference makes.  The later steps of ``_init_search_graph`` -- hub search tree, reordering of data and
graph by tree leaf order (1629-1651) -- belong to the query path and are out of scope.
"""
import numpy as np
import scipy.sparse as sp

from . import _capi

FLOAT32_EPS = np.finfo(np.float32).eps  # pynndescent_.py:65


def build_search_graph(data, indices, distances, metric="euclidean", n_neighbors=None, pruning_degree_multiplier=1.5,
                       diversify_prob=1.0, diversify_method="standard", device=0, return_stages=False):
    """(indices int32 (n,k), alt-space distances float32 (n,k)) -> scipy CSR uint8 search graph (unordered).

    ``indices``/``distances`` are ``NNDescent._neighbor_graph`` (rows ascending, squared-L2 / log2-cosine)."""
    if diversify_method != "standard" or diversify_prob != 1.0:
        raise NotImplementedError("only the reference defaults diversify_method='standard', diversify_prob=1.0 run on the GPU")
    x = np.ascontiguousarray(data, dtype=np.float32)
    n, d = x.shape
    k = indices.shape[1]
    n_neighbors = k if n_neighbors is None else n_neighbors
    code = {"euclidean": _capi.NND_METRIC_SQEUCLIDEAN, "l2": _capi.NND_METRIC_SQEUCLIDEAN, "cosine": _capi.NND_METRIC_ALT_COSINE}[metric]
    b = _capi.Builder(n, d, code, k, 0, 60, 200, min(60, k), 1, 0.001, [1, 2, 3], [4, 5, 6], device=device)
    try:
        b.set_data_host(x)
        rows, dd = b.diversify(indices, distances)  # pynndescent_.py:1502-1511
        nnz_pre = int((np.asarray(indices) >= 0).sum())
        dd[dd == 0.0] = FLOAT32_EPS  # preserve distance-0 points (pynndescent_.py:1517)
        # COO -> CSR (1520-1527): entries stay in row order (ascending distance), -1 slots dropped
        keep = rows >= 0
        indptr = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int32)
        fwd = sp.csr_array((dd[keep].astype(np.float32), rows[keep].astype(np.int32), indptr), shape=(n, n))
        # "Reverse graph" (1541-1577): scipy's transpose of a CSR matrix is a CSC view of the SAME arrays, so
        # diversify_csr sees the forward rows; the surviving weights are then read as the transposed matrix.
        rdata = b.diversify_csr(fwd.indptr, fwd.indices, fwd.data)
        rev = sp.csr_array((rdata, fwd.indices.copy(), fwd.indptr.copy()), shape=(n, n)).transpose().tocsr()
        rev.eliminate_zeros()
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
        return graph, {"forward_rows": rows, "forward_dist": dd, "nnz_pre_diversify": nnz_pre, "forward_nnz": int(fwd.nnz),
                       "reverse_nnz": int(rev.nnz), "union_nnz": nnz_pre_prune, "final_nnz": int(graph.nnz)}
    return graph

"""The graph-informed ("hub") search tree of ``NNDescent.prepare()`` and the reordering by its leaf order.

Mirrors ``make_hub_tree`` + ``convert_tree_format`` (reference rp_trees.py:714-1312, 2926-3049) and the tail of
``_init_search_graph`` (pynndescent_.py:1436-1449, 1629-1651).  The tree is built on the GPU (csrc/hubtree.hip, entry
points ``nnd_hub_tree_build`` / ``nnd_hub_tree_fetch``); what stays on the host is what the reference does with
numpy / scipy too: the in-degree bincount, the (-degree, id) order, fancy-indexing the data and the CSR graph.
"""
from collections import namedtuple

import numpy as np

from . import _capi

# the reference's FlatTree (rp_trees.py:27-29); a namedtuple with the same field names, so code written against the
# reference's attribute names (and its pickle layout, rp_trees.py:3052-3081) works on it unchanged
FlatTree = namedtuple("FlatTree", ["hyperplanes", "offsets", "children", "indices", "leaf_size"])


def compute_global_degrees(neighbor_indices):
    """rp_trees.py:714-744: how often every point appears as somebody's neighbour."""
    idx = np.asarray(neighbor_indices)
    n = idx.shape[0]
    valid = (idx >= 0) & (idx < n)
    return np.bincount(idx[valid].ravel(), minlength=n)[:n].astype(np.int32)


def make_hub_tree(data, neighbor_indices, metric="euclidean", leaf_size=30, max_depth=200, device=0, seed=0):
    """FlatTree of the hub search tree for ``data`` (float32 (n, d)) and its k-NN graph ``neighbor_indices`` (n, k).

    Hubs of a node = its members of highest in-degree, ties to the smaller id (get_top_k_hub_indices keeps the member
    that comes first, rp_trees.py:747-798, and members stay in id order): a global order by (-degree, id)."""
    x = np.ascontiguousarray(data, np.float32)
    n, d = x.shape
    deg = compute_global_degrees(neighbor_indices)
    rank_order = np.argsort(-deg.astype(np.int64), kind="stable").astype(np.int32)
    code = _capi.METRIC_CODES[metric]
    b = _capi.Builder(n, d, code, 1, 1, max(int(leaf_size), 1), max_depth, 1, 1, 0.001, [seed, 2, 3], [4, 5, 6], device=device,
                      flags=_capi.NND_FLAG_NO_GRAPH | _capi.NND_FLAG_NO_PREP)  # original rows + the forest's scan / scatter buffers
    try:
        b.set_data_host(x)
        hyper, offs, children, indices, leaf = b.hub_tree(rank_order, leaf_size, max_depth)
    finally:
        b.close()
    return FlatTree(hyper, offs, children, indices, leaf)


def search_flat_tree(tree, points, rng=None):
    """Leaf bounds (start, end) in ``tree.indices`` for every row of ``points`` (select_side / search_flat_tree,
    rp_trees.py:2662-2741), vectorised over the batch; exact ties (|margin| < 1e-8) are broken by ``rng``."""
    pts = np.ascontiguousarray(points, np.float32)
    node = np.zeros(pts.shape[0], np.int64)
    active = tree.children[node, 0] > 0
    rng = np.random.RandomState(0) if rng is None else rng
    while active.any():
        a = np.nonzero(active)[0]
        nd = node[a]
        margin = tree.offsets[nd] + np.einsum("ij,ij->i", tree.hyperplanes[nd], pts[a])
        side = np.where(np.abs(margin) < 1e-8, rng.randint(0, 2, a.shape[0]), (margin <= 0).astype(np.int64))
        node[a] = tree.children[nd, side]
        active = tree.children[node, 0] > 0
    return -tree.children[node, 0], -tree.children[node, 1]


def reorder_by_tree(search_graph, raw_data, tree):
    """pynndescent_.py:1629-1651: rows and columns of the search graph and the data rows in the tree's leaf order; the
    tree's own index array becomes the identity.  Returns (graph, data, vertex_order, tree)."""
    vertex_order = np.asarray(tree.indices)
    g = search_graph[vertex_order, :].tocsc()
    g = g[:, vertex_order].tocsr()
    g.sort_indices()
    data = np.ascontiguousarray(raw_data[vertex_order, :])
    tree_order = np.argsort(vertex_order)
    new_tree = FlatTree(tree.hyperplanes, tree.offsets, tree.children, tree.indices[tree_order].astype(np.int32, order="C"), tree.leaf_size)
    return g, data, vertex_order, new_tree

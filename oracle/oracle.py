"""ctypes binding of oracle/nnd_oracle.c -- TEST INFRASTRUCTURE ONLY.

Mirrors the reference call sequence of ``NNDescent.__init__`` for the dense
euclidean / cosine branch (reference pynndescent_.py:1105-1133, 1247-1260):
RandomState draws -> make_forest -> rptree_leaf_array -> nn_descent.
The product path (pynndescent_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
INT32_MIN = np.iinfo(np.int32).min + 1  # reference pynndescent_.py:62
INT32_MAX = np.iinfo(np.int32).max - 1  # reference pynndescent_.py:63
METRICS = {"euclidean": 0, "l2": 0, "cosine": 1}

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")


class Trace(C.Structure):
    _fields_ = [
        ("n_iters_run", C.c_int64),
        ("c", C.c_int64 * 64),
        ("pairs", C.c_int64 * 64),
        ("generated", C.c_int64 * 64),
    ]


def build(force=False):
    """Compile both oracle libraries (gcc); no-op when they are up to date."""
    out = os.path.join(_HERE, "_build")
    src = os.path.join(_HERE, "nnd_oracle.c")
    libs = [os.path.join(out, "liboracle_strict.so"), os.path.join(out, "liboracle_fast.so")]
    stale = force or any(
        (not os.path.exists(p)) or os.path.getmtime(p) < os.path.getmtime(src) for p in libs
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])
    return libs


def build_native():
    """``liboracle_native.so``: the 'fast' flags with -march=native, compiled on the box that runs it (bench.py's cpu_baseline leg
    on the GPU box's host; the library is never committed or shipped).  Returns 'native' when it could be built and loaded, else 'fast'."""
    path = os.path.join(_HERE, "_build", "liboracle_native.so")
    try:
        import hashlib

        src = os.path.join(_HERE, "nnd_oracle.c")
        with open("/proc/cpuinfo") as fh:  # the library is only valid on the CPU it was compiled on: a copy that travelled is rebuilt
            cpu = [ln for ln in fh.read().splitlines() if ln.startswith(("model name", "flags"))][:2]
        tag = hashlib.sha1("\n".join(cpu).encode()).hexdigest()
        stamp = path + ".cpu"
        same = os.path.exists(stamp) and open(stamp).read().strip() == tag
        if (not os.path.exists(path)) or not same or os.path.getmtime(path) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            with open(stamp, "w") as fh:
                fh.write(tag)
        load("native")
        return "native"
    except Exception:
        return "fast"


_LIBS = {}


def load(kind="strict"):
    """Load ``liboracle_<kind>.so`` (kind: 'strict' for parity pins, 'fast' for timing)."""
    if kind in _LIBS:
        return _LIBS[kind]
    path = os.path.join(_HERE, "_build", "liboracle_%s.so" % kind)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.orc_tau_rand_int.argtypes = [_i64p]
    lib.orc_tau_rand_int.restype = C.c_int32
    lib.orc_tau_rand.argtypes = [_i64p]
    lib.orc_tau_rand.restype = C.c_float
    for name in ("orc_squared_euclidean", "orc_alternative_cosine"):
        fn = getattr(lib, name)
        fn.argtypes = [_f32p, _f32p, C.c_int]
        fn.restype = C.c_float
    for name in ("orc_euclidean", "orc_cosine"):
        fn = getattr(lib, name)
        fn.argtypes = [_f32p, _f32p, C.c_int]
        fn.restype = C.c_double
    lib.orc_correct_distances.argtypes = [_f32p, _f64p, C.c_int64, C.c_int]
    lib.orc_make_heap.argtypes = [_i32p, _f32p, _u8p, C.c_int64, C.c_int]
    lib.orc_checked_flagged_heap_push.argtypes = [_f32p, _i32p, _u8p, C.c_int, C.c_float, C.c_int32, C.c_uint8]
    lib.orc_checked_flagged_heap_push.restype = C.c_int
    lib.orc_checked_heap_push.argtypes = [_f32p, _i32p, C.c_int, C.c_float, C.c_int32]
    lib.orc_checked_heap_push.restype = C.c_int
    lib.orc_deheap_sort.argtypes = [_i32p, _f32p, C.c_int64, C.c_int]
    for name in ("orc_euclidean_split", "orc_angular_split"):
        fn = getattr(lib, name)
        fn.argtypes = [_f32p, C.c_int, _i32p, C.c_int, _i64p, _i8p, _f32p, C.POINTER(C.c_float)]
        fn.restype = C.c_int
    lib.orc_make_forest_leaf_array.argtypes = [
        _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, _i64p, C.c_int, C.c_int,
        C.POINTER(C.c_int64), C.POINTER(C.c_int32),
    ]
    lib.orc_make_forest_leaf_array.restype = C.POINTER(C.c_int32)
    lib.orc_free.argtypes = [C.c_void_p]
    lib.orc_init_rp_tree.argtypes = [
        _f32p, C.c_int64, C.c_int, C.c_int, _i32p, _f32p, _u8p, C.c_int, _i32p, C.c_int64, C.c_int, C.c_int,
    ]
    lib.orc_init_random.argtypes = [_f32p, C.c_int64, C.c_int, C.c_int, _i32p, _f32p, _u8p, C.c_int, _i64p]
    lib.orc_init_from_graph.argtypes = [
        _f32p, C.c_int64, C.c_int, C.c_int, _i32p, _f32p, _u8p, C.c_int, _i32p, C.c_void_p, C.c_int,
    ]
    lib.orc_init_from_neighbor_graph.argtypes = [_i32p, _f32p, _u8p, C.c_int, _i32p, _f32p, C.c_int64, C.c_int]
    lib.orc_new_build_candidates.argtypes = [
        _i32p, _u8p, C.c_int64, C.c_int, C.c_int, _i64p, C.c_int, _i32p, _i32p,
    ]
    lib.orc_nn_descent_internal.argtypes = [
        _f32p, C.c_int64, C.c_int, C.c_int, _i32p, _f32p, _u8p, C.c_int, _i64p, C.c_int, C.c_int, C.c_float,
        C.c_int, C.POINTER(Trace),
    ]
    lib.orc_nn_descent.argtypes = [
        _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, _i64p, C.c_int, C.c_int, C.c_float, _i32p, C.c_int64,
        C.c_int, _i32p, _f32p, _u8p, C.c_int, C.c_int, C.POINTER(Trace),
    ]
    lib.orc_nn_descent.restype = C.c_int
    lib.orc_brute_force_knn.argtypes = [
        _f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, _i32p, _f64p,
    ]
    lib.orc_num_threads.restype = C.c_int
    lib.orc_diversify.argtypes = [_i32p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, C.c_int]
    lib.orc_diversify_csr.argtypes = [_i32p, _i32p, _f32p, C.c_int64, _f32p, C.c_int, C.c_int]
    lib.orc_degree_prune.argtypes = [_i32p, _f32p, C.c_int64, C.c_int]
    lib.orc_diversify_p.argtypes = [_i32p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, C.c_int, _i64p, C.c_float]
    lib.orc_diversify_degree_aware.argtypes = [_i32p, _f32p, C.c_int64, C.c_int, _f32p, C.c_int, C.c_int, C.c_int,
                                               C.c_float, C.c_float]
    lib.orc_diversify_csr_p.argtypes = [_i32p, _i32p, _f32p, C.c_int64, _f32p, C.c_int, C.c_int, _i64p, C.c_float]
    lib.orc_diversify_csr_degree_aware.argtypes = [_i32p, _i32p, _f32p, C.c_int64, _f32p, C.c_int, C.c_int, _i64p, C.c_int,
                                                   C.c_float, C.c_float]
    lib.orc_make_hub_tree.argtypes = [_f32p, C.c_int64, C.c_int, _i32p, C.c_int, _i64p, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)),
                                      C.POINTER(C.POINTER(C.c_int32)), _i32p, C.POINTER(C.c_int32)]
    lib.orc_make_hub_tree.restype = C.c_int64
    _LIBS[kind] = lib
    return lib


# ----------------------------------------------------------------------------
# reference-shaped helpers


def default_n_trees(n):  # reference pynndescent_.py:1009-1010
    return max(3, min(12, int(round(2.0 * np.log10(n)))))


def default_n_iters(n):  # reference pynndescent_.py:1011-1012
    return max(5, int(round(np.log2(n))))


def default_leaf_size(n_neighbors):  # reference rp_trees.py:2845-2846
    return max(60, min(256, 5 * int(n_neighbors)))


def draw_rng_states(random_state, n_trees, tree_init=True):
    """The three RandomState draws of the build, in reference order
    (pynndescent_.py:1105-1110, rp_trees.py:2850)."""
    rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    rng_state = rs.randint(INT32_MIN, INT32_MAX, 3).astype(np.int64)
    search_rng_state = rs.randint(INT32_MIN, INT32_MAX, 3).astype(np.int64)
    tree_states = (
        rs.randint(INT32_MIN, INT32_MAX, size=(n_trees, 3)).astype(np.int64)
        if (tree_init and n_trees > 0)
        else np.zeros((0, 3), np.int64)
    )
    return rng_state, search_rng_state, tree_states


def make_leaf_array(data, n_trees, leaf_size, tree_states, angular, max_depth=200, lib=None):
    """make_forest + rptree_leaf_array (reference rp_trees.py:2815-2922)."""
    lib = lib or load()
    data = np.ascontiguousarray(data, np.float32)
    if n_trees == 0:
        return np.array([[-1]], dtype=np.int32)
    nl = C.c_int64()
    ms = C.c_int32()
    ptr = lib.orc_make_forest_leaf_array(
        data, data.shape[0], data.shape[1], n_trees, leaf_size,
        np.ascontiguousarray(tree_states, np.int64), int(bool(angular)), max_depth, C.byref(nl), C.byref(ms),
    )
    arr = np.ctypeslib.as_array(ptr, shape=(max(nl.value, 1), ms.value)).copy()[: nl.value]
    lib.orc_free(ptr)
    return arr


def init_rp_tree(data, n_neighbors, metric, leaf_array, lib=None):
    """make_heap + init_rp_tree (reference pynndescent_.py:116-185, utils.py:130-158) on a given leaf array.
    Returns the heap triple (indices, distances, flags), heap-ordered rows."""
    lib = lib or load()
    data = np.ascontiguousarray(data, np.float32)
    n, dim = data.shape
    k = int(n_neighbors)
    hi = np.empty((n, k), np.int32)
    hd = np.empty((n, k), np.float32)
    hf = np.empty((n, k), np.uint8)
    lib.orc_make_heap(hi, hd, hf, n, k)
    la = np.ascontiguousarray(leaf_array, np.int32)
    lib.orc_init_rp_tree(data, n, dim, METRICS[metric], hi, hd, hf, k, la, la.shape[0], la.shape[1], 8)
    return hi, hd, hf


def nn_descent(data, n_neighbors, rng_state, max_candidates, metric, n_iters, delta, leaf_array,
               n_threads=8, init=None, lib=None, return_trace=False):
    """reference pynndescent_.py:323-366. Returns (indices, alt-space distances), rows ascending."""
    lib = lib or load()
    data = np.ascontiguousarray(data, np.float32)
    n, dim = data.shape
    k = int(n_neighbors)
    if init is None:
        hi = np.empty((n, k), np.int32)
        hd = np.empty((n, k), np.float32)
        hf = np.empty((n, k), np.uint8)
        have = 0
    else:
        hi, hd, hf = (np.ascontiguousarray(a).copy() for a in init)
        have = 1
    la = np.ascontiguousarray(leaf_array, np.int32)
    tr = Trace()
    st = np.ascontiguousarray(rng_state, np.int64)
    lib.orc_nn_descent(data, n, dim, METRICS[metric], k, st, int(max_candidates), int(n_iters), float(delta),
                       la, la.shape[0], la.shape[1], hi, hd, hf, have, int(n_threads), C.byref(tr))
    if return_trace:
        it = tr.n_iters_run
        return hi, hd, {"iters": it, "c": list(tr.c[:it]), "pairs": list(tr.pairs[:it]),
                        "generated": list(tr.generated[:it])}
    return hi, hd


def build_index(data, metric="euclidean", n_neighbors=30, n_trees=None, leaf_size=None, random_state=None,
                max_candidates=None, n_iters=None, delta=0.001, tree_init=True, max_rptree_depth=200,
                n_threads=8, kind="strict", return_trace=False):
    """The dense build of ``NNDescent.__init__`` (reference pynndescent_.py:976-1260), CPU oracle.
    Returns (indices int32 (n,k), alt-space distances f32 (n,k))[, trace]."""
    lib = load(kind)
    data = np.ascontiguousarray(data, np.float32)
    n = data.shape[0]
    if n_trees is None:
        n_trees = default_n_trees(n)
    if n_iters is None:
        n_iters = default_n_iters(n)
    tree_init = bool(tree_init) and n_trees > 0
    rng_state, _, tree_states = draw_rng_states(random_state, n_trees, tree_init)
    if tree_init:
        ls = default_leaf_size(n_neighbors) if leaf_size is None else leaf_size
        leaf_array = make_leaf_array(data, n_trees, ls, tree_states, metric == "cosine", max_rptree_depth, lib)
    else:
        leaf_array = np.array([[-1]], dtype=np.int32)
    mc = min(60, n_neighbors) if max_candidates is None else max_candidates  # pynndescent_.py:1135-1138
    return nn_descent(data, n_neighbors, rng_state, mc, metric, n_iters, delta, leaf_array,
                      n_threads=n_threads, lib=lib, return_trace=return_trace)


def update_index(raw_data, graph, rng_state, random_state, metric="euclidean", n_neighbors=30, n_trees_after_update=2,
                 leaf_size=None, max_candidates=None, n_iters=None, delta=0.001, max_rptree_depth=200,
                 xs_fresh=None, xs_updated=None, updated_indices=None, n_threads=8, kind="strict"):
    """``NNDescent.update`` (reference pynndescent_.py:2381-2553) for a dense index that has not been prepared.

    raw_data: the index's current data; graph: its (indices, alt-space distances); rng_state: the index's
    ``rng_state`` array AS LEFT BY THE BUILD (init_random advanced it in place); random_state: the index's
    ``random_state`` (a RandomState object continues its stream, an int restarts it -- check_random_state).
    Returns (new raw data, (indices, alt-space distances))."""
    lib = load(kind)
    rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    _unused = rs.randint(INT32_MIN, INT32_MAX, 3).astype(np.int64)  # pynndescent_.py:2408-2411 (handed to make_forest, unused there)
    raw = np.array(raw_data, np.float32, copy=True)
    gi = np.array(graph[0], np.int32, copy=True)
    gd = np.array(graph[1], np.float32, copy=True)
    upd = [] if updated_indices is None else [int(i) for i in updated_indices]
    if xs_updated is not None:  # pynndescent_.py:2476-2493
        for row, i in zip(np.asarray(xs_updated, np.float32), upd):
            raw[i] = row
        hit = np.zeros(raw.shape[0], bool)
        hit[upd] = True
        gd[hit] = np.inf
        gi[hit] = -1
        stale = (gi >= 0) & hit[np.clip(gi, 0, None)]
        gi[stale] = -1
        gd[stale] = np.inf
    fresh = np.zeros((0, raw.shape[1]), np.float32) if xs_fresh is None else np.asarray(xs_fresh, np.float32)
    raw = np.ascontiguousarray(np.vstack([raw, fresh]))
    n, dim = raw.shape
    k = int(n_neighbors)
    tree_states = rs.randint(INT32_MIN, INT32_MAX, size=(n_trees_after_update, 3)).astype(np.int64)  # rp_trees.py:2850
    ls = default_leaf_size(k) if leaf_size is None else leaf_size
    leaf_array = make_leaf_array(raw, n_trees_after_update, ls, tree_states, metric == "cosine", max_rptree_depth, lib)
    hi = np.empty((n, k), np.int32)
    hd = np.empty((n, k), np.float32)
    hf = np.empty((n, k), np.uint8)
    lib.orc_make_heap(hi, hd, hf, n, k)
    lib.orc_init_from_neighbor_graph(hi, hd, hf, k, gi, gd, gi.shape[0], gi.shape[1])  # pynndescent_.py:2513-2516
    la = np.ascontiguousarray(leaf_array, np.int32)
    lib.orc_init_rp_tree(raw, n, dim, METRICS[metric], hi, hd, hf, k, la, la.shape[0], la.shape[1], 8)  # :2517
    mc = min(60, k) if max_candidates is None else max_candidates
    if n_iters is None:
        n_iters = default_n_iters(n)
    out = nn_descent(raw, k, rng_state, mc, metric, n_iters, delta, np.array([[-1], [-1]], np.int32),
                     n_threads=n_threads, init=(hi, hd, hf), lib=lib)
    return raw, out


def correct_distances(alt, metric):
    """``_distance_correction`` (reference distances.py:2170-2173, 704-711)."""
    lib = load()
    alt = np.ascontiguousarray(alt, np.float32)
    out = np.empty(alt.shape, np.float64)
    lib.orc_correct_distances(alt.reshape(-1), out.reshape(-1), alt.size, METRICS[metric])
    return out


def brute_force_knn(data, k, metric="euclidean", rows=None, kind="fast"):
    """Exact kNN, self included, true-metric float64 distances (ground truth T2)."""
    lib = load(kind)
    data = np.ascontiguousarray(data, np.float32)
    n, dim = data.shape
    if rows is None:
        nr, rp = n, None
    else:
        rows = np.ascontiguousarray(rows, np.int64)
        nr, rp = rows.shape[0], rows.ctypes.data_as(C.c_void_p)
    oi = np.empty((nr, k), np.int32)
    od = np.empty((nr, k), np.float64)
    lib.orc_brute_force_knn(data, n, dim, METRICS[metric], k, rp, nr, oi, od)
    return oi, od


def recall(true_idx, approx_idx, k_true=None):
    """Reference-test convention (tests/test_pynndescent_.py:27-31): fraction of the true
    top-``k_true`` found anywhere in the approximate row."""
    k_true = true_idx.shape[1] if k_true is None else k_true
    hits = 0
    for t, a in zip(true_idx[:, :k_true], approx_idx):
        hits += np.isin(t, a).sum()
    return hits / float(true_idx.shape[0] * k_true)


# ----------------------------------------------------------------------------
# search-graph pruning pass (BASELINE config 5): reference pynndescent_.py:1451-1611 without the tree reordering

FLOAT32_EPS = np.finfo(np.float32).eps


def diversify(indices, distances, data, metric, lib=None):
    """reference pynndescent_.py:369-403 (standard method, diversify_prob = 1). Returns new arrays."""
    lib = lib or load()
    i = np.ascontiguousarray(indices, np.int32).copy()
    d = np.ascontiguousarray(distances, np.float32).copy()
    x = np.ascontiguousarray(data, np.float32)
    lib.orc_diversify(i, d, i.shape[0], i.shape[1], x, x.shape[1], METRICS[metric])
    return i, d


def search_graph(data, indices, distances, metric, n_neighbors, pruning_degree_multiplier=1.5, lib=None,
                 return_stages=False, diversify_prob=1.0, diversify_method="standard", degree_prune_aggressiveness=1.0,
                 rng_state=None):
    """The pruning pass of ``NNDescent._init_search_graph`` (pynndescent_.py:1451-1611) on an (indices,
    alt-space distances) neighbour graph: forward diversify -> CSR -> reverse diversify -> union by maximum ->
    no diagonal -> degree prune to round(multiplier * k) -> binarise.  Vertex reordering by the search tree
    (pynndescent_.py:1629-1651) is NOT applied.  Returns a scipy CSR uint8 matrix.

    ``diversify_method`` 'standard' | 'degree_aware', ``diversify_prob`` and ``degree_prune_aggressiveness`` as in the
    reference constructor; ``rng_state`` (int64[3], advanced in place) feeds the coins when ``diversify_prob < 1``."""
    import scipy.sparse as sp

    lib = lib or load()
    x = np.ascontiguousarray(data, np.float32)
    n = x.shape[0]
    aware = diversify_method == "degree_aware"
    st = np.array([1, 2, 3], np.int64) if rng_state is None else rng_state
    if aware:  # pynndescent_.py:1476-1498 (diversify_prob is handed to `alpha`)
        rows = np.ascontiguousarray(indices, np.int32).copy()
        dd = np.ascontiguousarray(distances, np.float32).copy()
        lib.orc_diversify_degree_aware(rows, dd, n, rows.shape[1], x, x.shape[1], METRICS[metric],
                                       max(1, int(pruning_degree_multiplier * n_neighbors)),
                                       float(degree_prune_aggressiveness), float(diversify_prob))
    elif diversify_prob < 1.0:
        rows = np.ascontiguousarray(indices, np.int32).copy()
        dd = np.ascontiguousarray(distances, np.float32).copy()
        lib.orc_diversify_p(rows, dd, n, rows.shape[1], x, x.shape[1], METRICS[metric], st, float(diversify_prob))
    else:
        rows, dd = diversify(indices, distances, x, metric, lib)
    dd = dd.copy()
    dd[dd == 0.0] = FLOAT32_EPS  # pynndescent_.py:1525
    # COO -> CSR as scipy builds it (pynndescent_.py:1527-1537): entries stay in row order, i.e. ascending
    # distance, -1 slots dropped; the forward graph is NOT index-sorted at this point
    keep = rows >= 0
    indptr = np.concatenate([[0], np.cumsum(keep.sum(1))]).astype(np.int32)
    f_indices = np.ascontiguousarray(rows[keep], np.int32)
    # "Reverse graph" (pynndescent_.py:1549-1588): scipy's transpose of a CSR matrix is a CSC matrix over the SAME
    # (indptr, indices, data) arrays, so what the reference hands to diversify_csr are the forward rows again; and
    # because the arrays are shared, reverse_graph.eliminate_zeros() (in place) removes the pruned edges from the
    # FORWARD matrix too.  The union below is therefore max(F', F'^T) of the doubly pruned F'.
    rdata = np.ascontiguousarray(dd[keep], np.float32).copy()
    if aware:
        lib.orc_diversify_csr_degree_aware(indptr, f_indices, rdata, n, x, x.shape[1], METRICS[metric], st,
                                           int(n_neighbors), float(degree_prune_aggressiveness), float(diversify_prob))
    elif diversify_prob < 1.0:
        lib.orc_diversify_csr_p(indptr, f_indices, rdata, n, x, x.shape[1], METRICS[metric], st, float(diversify_prob))
    else:
        lib.orc_diversify_csr(indptr, f_indices, rdata, n, x, x.shape[1], METRICS[metric])
    fwd = sp.csr_array((rdata, f_indices, indptr), shape=(n, n))
    fwd.eliminate_zeros()
    rev = fwd.transpose().tocsr()
    rev.sort_indices()
    fwd.sort_indices()
    u = fwd.maximum(rev).tocsr()
    u.setdiag(0.0)
    u.eliminate_zeros()
    udata = np.ascontiguousarray(u.data, np.float32)
    lib.orc_degree_prune(np.ascontiguousarray(u.indptr, np.int32), udata, n,
                         int(np.round(pruning_degree_multiplier * n_neighbors)))
    u = sp.csr_array((udata, u.indices, u.indptr), shape=(n, n))
    u.eliminate_zeros()
    out = (u != 0).astype(np.uint8).tocsr()
    out.sort_indices()
    if return_stages:
        return out, {"forward_rows": rows, "forward_dist": dd, "reverse_nnz": int(rev.nnz), "union_nnz": int(u.nnz)}
    return out


# ----------------------------------------------------------------------------
# hub search tree of NNDescent.prepare (reference rp_trees.py:714-1312 + convert_tree_format rp_trees.py:2926-3049)

def make_hub_tree(data, neighbor_indices, rng_state, leaf_size=30, angular=False, max_depth=200, lib=None):
    """Returns the FlatTree fields (hyperplanes (n_nodes, dim), offsets, children (n_nodes, 2), indices (n), leaf_size)."""
    lib = lib or load()
    x = np.ascontiguousarray(data, np.float32)
    nb = np.ascontiguousarray(neighbor_indices, np.int32)
    n, dim = x.shape
    st = np.ascontiguousarray(rng_state, np.int64).copy()
    ph, po, pc = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_int32)()
    indices = np.empty(n, np.int32)
    ml = C.c_int32()
    nn = lib.orc_make_hub_tree(x, n, dim, nb, nb.shape[1], st, int(leaf_size), int(bool(angular)), int(max_depth),
                               C.byref(ph), C.byref(po), C.byref(pc), indices, C.byref(ml))
    hyper = np.ctypeslib.as_array(ph, shape=(nn, dim)).copy()
    offs = np.ctypeslib.as_array(po, shape=(nn,)).copy()
    children = np.ctypeslib.as_array(pc, shape=(nn, 2)).copy()
    for p_ in (ph, po, pc):
        lib.orc_free(C.cast(p_, C.c_void_p))
    return hyper, offs, children, indices, int(ml.value)

"""Generate the golden fixtures in this directory from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [--only NAME ...]

It imports the unmodified reference package through ``oracle/ref_t0.py`` (stub
numba, source executed un-jitted -- "T0") and records
inputs/outputs of the functions on the build hot path.  The fixtures pin the C
oracle (``oracle/nnd_oracle.c``); the GPU path is then compared with the oracle.
Nothing here is read by the product path, and nothing at test time reads
/root/reference.

Fixture inputs are either stored or regenerated from ``np.random.RandomState``
seeds recorded in the file (legacy generator: stable across numpy versions).
The three ``reference_testdata_*`` inputs are the data files of the reference's
own tests (pynndescent/tests/test_data), stored as compressed float32/int32.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_t0  # noqa: E402

REF_TESTDATA = os.path.join(ref_t0.REFERENCE_ROOT, "pynndescent", "tests", "test_data")


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024.0))


# ----------------------------------------------------------------------------
# synthetic inputs shared with the tests (tests/util_data.py re-implements these)


def clustered(n, d, latent, n_clusters, seed, noise=0.3, nonneg=False):
    """Low-intrinsic-dimension Gaussian mixture (SURVEY.md section 8d generator)."""
    rs = np.random.RandomState(seed)
    centres = rs.standard_normal((n_clusters, latent)) * 3.0
    assign = rs.randint(0, n_clusters, n)
    z = centres[assign] + rs.standard_normal((n, latent))
    proj = rs.standard_normal((latent, d)) / np.sqrt(latent)
    x = z @ proj + noise * rs.standard_normal((n, d))
    if nonneg:
        x = x - x.min()
    return np.ascontiguousarray(x, dtype=np.float32)


def nn_data_like(seed=189212):
    """Shape/convention of the reference fixture nn_data (tests/conftest.py:47-52)."""
    rs = np.random.RandomState(seed)
    x = rs.uniform(0, 1, size=(1000, 5))
    return np.vstack([x, np.zeros((2, 5))]).astype(np.float32)


# ----------------------------------------------------------------------------


def gen_primitives(ref):
    from pynndescent import distances as D
    from pynndescent import utils as U

    rs = np.random.RandomState(7)
    # tau_rand (utils.py:17-57)
    states0 = rs.randint(-(2**31) + 1, 2**31 - 2, size=(6, 3)).astype(np.int64)
    ints = np.zeros((6, 40), np.int32)
    floats = np.zeros((6, 40), np.float32)
    states_after = np.zeros_like(states0)
    for r in range(6):
        s = states0[r].copy()
        for c in range(40):
            ints[r, c] = U.tau_rand_int(s)
        s2 = states0[r].copy()
        for c in range(40):
            floats[r, c] = U.tau_rand(s2)
        states_after[r] = s
    # distances (distances.py:50-91, 555-630, 704-711)
    xs = rs.standard_normal((40, 33)).astype(np.float32)
    ys = rs.standard_normal((40, 33)).astype(np.float32)
    xs[3] = 0
    ys[3] = 0  # both zero
    xs[5] = 0  # one zero
    ys[7] = 0
    ys[9] = xs[9]  # identical
    ys[11] = -xs[11]  # opposite
    ys[13] = xs[13] * 3.0  # parallel
    sq = np.array([D.squared_euclidean(x, y) for x, y in zip(xs, ys)], np.float32)
    ac = np.array([D.alternative_cosine(x, y) for x, y in zip(xs, ys)], np.float32)
    eu = np.array([D.euclidean(x, y) for x, y in zip(xs, ys)], np.float64)
    co = np.array([D.cosine(x, y) for x, y in zip(xs, ys)], np.float64)
    corr_e = np.sqrt(sq)
    finite = ac < 1e30
    corr_c = np.asarray(D.correct_alternative_cosine(np.where(finite, ac, 0).astype(np.float32)), np.float64)
    # heap pushes (utils.py:409-533) + deheap_sort (utils.py:189-218)
    size = 10
    n_push = 80
    push_p = rs.uniform(0, 1, n_push).astype(np.float32)
    push_p[20] = push_p[10]  # a distance tie
    push_n = rs.randint(0, 25, n_push).astype(np.int32)  # small id range -> duplicates
    push_f = rs.randint(0, 2, n_push).astype(np.uint8)
    heap = U.make_heap(1, size)
    hi, hd, hf = heap[0][0], heap[1][0], heap[2][0]
    ret = np.zeros(n_push, np.int32)
    snap_i = np.zeros((n_push, size), np.int32)
    snap_d = np.zeros((n_push, size), np.float32)
    snap_f = np.zeros((n_push, size), np.uint8)
    for t in range(n_push):
        ret[t] = U.checked_flagged_heap_push(hd, hi, hf, push_p[t], push_n[t], push_f[t])
        snap_i[t], snap_d[t], snap_f[t] = hi, hd, hf
    # unflagged variant
    pri = np.full(size, np.inf, np.float32)
    ind = np.full(size, -1, np.int32)
    ret2 = np.zeros(n_push, np.int32)
    for t in range(n_push):
        ret2[t] = U.checked_heap_push(pri, ind, push_p[t], push_n[t])
    # deheap_sort on 5 partially filled heaps
    heaps = U.make_heap(5, size)
    for r in range(5):
        for t in range(4 + 3 * r):
            U.checked_flagged_heap_push(
                heaps[1][r], heaps[0][r], heaps[2][r], np.float32(rs.uniform()), np.int32(rs.randint(0, 1000)), np.uint8(1)
            )
    pre_i, pre_d = heaps[0].copy(), heaps[1].copy()
    si, sd = U.deheap_sort(heaps[0].copy(), heaps[1].copy())
    save(
        "primitives",
        states0=states0, ints=ints, floats=floats, states_after=states_after,
        xs=xs, ys=ys, sq=sq, ac=ac, eu=eu, co=co, corr_e=corr_e, corr_c=corr_c, corr_c_mask=finite,
        push_p=push_p, push_n=push_n, push_f=push_f, push_ret=ret, snap_i=snap_i, snap_d=snap_d, snap_f=snap_f,
        push_ret_unflagged=ret2, unflagged_pri=pri, unflagged_ind=ind,
        sort_pre_i=pre_i, sort_pre_d=pre_d, sort_i=si, sort_d=sd,
    )


def gen_rp(ref):
    from pynndescent import rp_trees as R

    rs = np.random.RandomState(11)
    data = rs.standard_normal((300, 12)).astype(np.float32)
    data[10] = data[11]  # duplicate pair
    data[20:24] = 0.0  # zero rows
    indices = rs.permutation(300)[:200].astype(np.int32)
    out = {}
    for name, fn in (("euclid", R.euclidean_random_projection_split), ("angular", R.angular_random_projection_split)):
        for trial in range(3):
            st = rs.randint(-(2**31) + 1, 2**31 - 2, 3).astype(np.int64)
            st_in = st.copy()
            left, right, hyper, off = fn(data, indices, st)
            out["%s_%d_state_in" % (name, trial)] = st_in
            out["%s_%d_state_out" % (name, trial)] = st.copy()
            out["%s_%d_left" % (name, trial)] = np.asarray(left, np.int32)
            out["%s_%d_right" % (name, trial)] = np.asarray(right, np.int32)
            out["%s_%d_hyper" % (name, trial)] = np.asarray(hyper, np.float32)
            out["%s_%d_off" % (name, trial)] = np.float64(off)
    # degenerate: all points identical -> every margin is 0 -> coin flips (rp_trees.py:380-403)
    same = np.tile(rs.standard_normal((1, 6)).astype(np.float32), (50, 1))
    ind50 = np.arange(50, dtype=np.int32)
    st = np.array([12345, 67890, 13579], np.int64)
    left, right, hyper, off = R.euclidean_random_projection_split(same, ind50, st.copy())
    out["degenerate_left"] = np.asarray(left, np.int32)
    out["degenerate_right"] = np.asarray(right, np.int32)
    out["degenerate_data"] = same
    # forests -> leaf arrays (rp_trees.py:2815-2922)
    fdata = rs.uniform(0, 1, (700, 6)).astype(np.float32)
    for name, angular in (("euclid", False), ("angular", True)):
        rstate = np.random.RandomState(5)
        # make_forest draws its per-tree seeds from random_state (rp_trees.py:2850)
        forest = R.make_forest(fdata, 10, 3, 25, None, rstate, 1, angular, False, max_depth=200)
        la = R.rptree_leaf_array(forest)
        out["forest_%s_leaf_array" % name] = np.asarray(la, np.int32)
    out["forest_data"] = fdata
    # depth-limited forest: leaves larger than leaf_size (rp_trees.py:2188, 2548-2551)
    rstate = np.random.RandomState(6)
    forest = R.make_forest(fdata, 10, 2, 25, None, rstate, 1, False, False, max_depth=3)
    out["forest_depth3_leaf_array"] = np.asarray(R.rptree_leaf_array(forest), np.int32)
    save("rp", data=data, indices=indices, **out)


def gen_candidates(ref):
    import numba
    from pynndescent import utils as U

    rs = np.random.RandomState(13)
    n, k, mc = 400, 12, 8
    # a plausible heap state: random neighbours, ~half the rows partially filled, random flags
    heap = U.make_heap(n, k)
    for i in range(n):
        fill = k if i % 3 else rs.randint(0, k)
        for _ in range(fill + 4):
            U.checked_flagged_heap_push(
                heap[1][i], heap[0][i], heap[2][i], np.float32(rs.uniform()), np.int32(rs.randint(0, n)), np.uint8(rs.randint(0, 2))
            )
    out = {"heap_i": heap[0].copy(), "heap_d": heap[1].copy(), "heap_f": heap[2].copy()}
    rng_state = np.array([-1234567, 7654321, 424242], np.int64)
    for T in (1, 3):
        numba.set_num_threads(T)
        h = (heap[0].copy(), heap[1].copy(), heap[2].copy())
        new, old = U.new_build_candidates(h, mc, rng_state.copy(), T)
        out["T%d_new" % T] = np.asarray(new, np.int32)
        out["T%d_old" % T] = np.asarray(old, np.int32)
        out["T%d_flags_after" % T] = h[2].copy()
    numba.set_num_threads(1)
    save("candidates", rng_state=rng_state, mc=np.int32(mc), **out)


def _staged_build(ref, data, metric, k, n_trees, leaf_size, seed, n_iters, n_threads, tree_init=True,
                  max_candidates=None, delta=0.001):
    """The reference build, stage by stage, with the same calls NNDescent.__init__ makes
    (pynndescent_.py:1105-1133, 1247-1260; nn_descent pynndescent_.py:323-366), recording
    the graph after initialisation and the update count c of every iteration."""
    import numba
    from pynndescent import distances as D
    from pynndescent import pynndescent_ as P
    from pynndescent import rp_trees as R
    from pynndescent import utils as U

    numba.set_num_threads(n_threads)
    n = data.shape[0]
    rs = np.random.RandomState(seed)
    rng_state = rs.randint(P.INT32_MIN, P.INT32_MAX, 3).astype(np.int64)
    _search = rs.randint(P.INT32_MIN, P.INT32_MAX, 3).astype(np.int64)
    dist = D.fast_distance_alternatives[metric]["dist"]
    angular = metric == "cosine"
    if tree_init and n_trees > 0:
        forest = R.make_forest(data, k, n_trees, leaf_size, rng_state, rs, 1, angular, False, max_depth=200)
        leaf_array = R.rptree_leaf_array(forest)
    else:
        leaf_array = np.array([[-1]])
    mc = min(60, k) if max_candidates is None else max_candidates
    graph = U.make_heap(n, k)
    P.init_rp_tree(data, dist, graph, leaf_array)
    after_tree = (graph[0].copy(), graph[1].copy())
    P.init_random(k, data, graph, dist, rng_state)
    after_init = (graph[0].copy(), graph[1].copy(), graph[2].copy())
    # nn_descent_internal, unrolled (pynndescent_.py:266-320)
    block_size = 16384
    n_blocks = n // block_size
    T = numba.get_num_threads()
    max_updates = int((mc**2 + mc * (mc - 1) / 2) * block_size / T) + 1024
    update_array = np.empty((T, max_updates, 3), dtype=np.float32)
    n_upd = np.zeros(T, dtype=np.int32)
    cs = []
    for it in range(n_iters):
        new_c, old_c = U.new_build_candidates(graph, mc, rng_state, T)
        c = P.process_candidates(data, dist, graph, new_c, old_c, n_blocks, block_size, T, update_array, n_upd)
        cs.append(int(c))
        if c <= delta * k * n:
            break
    idx, dst = U.deheap_sort(graph[0], graph[1])
    numba.set_num_threads(1)
    return {
        "leaf_array": np.asarray(leaf_array, np.int32),
        "after_tree_idx": after_tree[0], "after_tree_dist": after_tree[1],
        "after_init_idx": after_init[0], "after_init_dist": after_init[1], "after_init_flags": after_init[2],
        "c": np.asarray(cs, np.int64),
        "idx": np.asarray(idx, np.int32), "dist": np.asarray(dst, np.float32),
        "rng_state_final": rng_state.copy(),
    }


def _class_build(ref, data, metric, k, seed, n_threads, **kw):
    """The reference class end to end (what a user calls)."""
    import numba

    numba.set_num_threads(n_threads)
    t0 = time.time()
    index = ref.NNDescent(data, metric=metric, n_neighbors=k, random_state=np.random.RandomState(seed), **kw)
    secs = time.time() - t0
    idx, dst = index._neighbor_graph
    cidx, cdst = index.neighbor_graph
    numba.set_num_threads(1)
    return {"idx": np.asarray(idx, np.int32), "dist": np.asarray(dst, np.float32),
            "corrected": np.asarray(cdst, np.float64), "t0_seconds": np.float64(secs)}


def gen_build_small(ref):
    # (a) the reference-test shape: 1002 x 5 uniform + 2 zero rows, default k=30 (tests/test_pynndescent_.py:19-53)
    x = nn_data_like()
    for metric in ("euclidean", "cosine"):
        for T in (1, 4):
            t0 = time.time()
            r = _staged_build(ref, x, metric, 30, 6, None, 189212, 10, T)
            print("nn_data", metric, T, "%.1fs" % (time.time() - t0), "c:", r["c"])
            save("build_nndata_%s_T%d" % (metric, T), seed=np.int64(189212), k=np.int32(30), n_trees=np.int32(6),
                 n_iters=np.int32(10), n_threads=np.int32(T), **r)
    # (b) class-level run with all defaults, as the reference tests construct it
    r = _class_build(ref, x, "euclidean", 30, 189212, 1)
    save("class_nndata_euclidean", seed=np.int64(189212), k=np.int32(30), **r)
    r = _class_build(ref, x, "cosine", 30, 189212, 1)
    save("class_nndata_cosine", seed=np.int64(189212), k=np.int32(30), **r)
    # (c) tree_init=False (tests/test_pynndescent_.py:666-684), leaf_size > n (716-747)
    r = _class_build(ref, x[200:], "euclidean", 10, 3, 1, tree_init=False)
    save("class_notree_euclidean", seed=np.int64(3), k=np.int32(10), **r)
    small = np.random.RandomState(17).uniform(5, 40, size=(10, 5)).astype(np.float32)
    r = _class_build(ref, small, "euclidean", 9, 4, 1, leaf_size=21)
    save("class_nosplit_euclidean", data=small, seed=np.int64(4), k=np.int32(9), leaf_size=np.int32(21), **r)


def gen_build_clustered(ref):
    x = clustered(3000, 16, 6, 40, seed=21)
    t0 = time.time()
    r = _staged_build(ref, x, "euclidean", 15, 8, None, 5, 12, 4)
    print("clustered euclid %.1fs" % (time.time() - t0), "c:", r["c"])
    save("build_clustered_euclidean_T4", gen=np.array([3000, 16, 6, 40, 21]), seed=np.int64(5), k=np.int32(15),
         n_trees=np.int32(8), n_iters=np.int32(12), n_threads=np.int32(4), **r)
    y = np.random.RandomState(22).standard_normal((2000, 24)).astype(np.float32)
    t0 = time.time()
    r = _staged_build(ref, y, "cosine", 15, 8, None, 6, 11, 4)
    print("iid cosine %.1fs" % (time.time() - t0), "c:", r["c"])
    save("build_iid_cosine_T4", gen=np.array([2000, 24, 22]), seed=np.int64(6), k=np.int32(15), n_trees=np.int32(8),
         n_iters=np.int32(11), n_threads=np.int32(4), **r)


def gen_build_wide(ref):
    """Rows of more than 64 neighbours and candidate lists of more than 64 entries (round 5 lifted the GPU path's bounds to 256 / 128;
    the reference has none, utils.py:130-158, pynndescent_.py:1135-1138)."""
    x = clustered(400, 10, 4, 8, seed=33)
    t0 = time.time()
    r = _staged_build(ref, x, "euclidean", 70, 2, None, 9, 4, 2, max_candidates=80)
    print("wide rows / lists %.1fs" % (time.time() - t0), "c:", r["c"])
    save("build_wide_euclidean_T2", gen=np.array([400, 10, 4, 8, 33]), seed=np.int64(9), k=np.int32(70), n_trees=np.int32(2),
         n_iters=np.int32(4), n_threads=np.int32(2), max_candidates=np.int32(80), **r)


def gen_build_c1(ref):
    """BASELINE.json configs[0]: 10k x 64 float32 random, euclidean, k=10, n_iters=5."""
    x = np.random.RandomState(0).standard_normal((10000, 64)).astype(np.float32)
    t0 = time.time()
    r = _staged_build(ref, x, "euclidean", 10, 8, None, 0, 5, 8)
    print("C1 %.1fs" % (time.time() - t0), "c:", r["c"])
    for drop in ("after_tree_dist", "after_init_dist", "after_init_flags", "leaf_array"):
        r.pop(drop)
    save("build_c1_T8", seed=np.int64(0), k=np.int32(10), n_trees=np.int32(8), n_iters=np.int32(5),
         n_threads=np.int32(8), **r)


def gen_reference_testdata(ref):
    hang = np.load(os.path.join(REF_TESTDATA, "cosine_hang.npy")).astype(np.float32)
    near = np.load(os.path.join(REF_TESTDATA, "cosine_near_duplicates.npy")).astype(np.float32)
    bug = np.load(os.path.join(REF_TESTDATA, "pynndescent_bug_np.npz"))["arr_0"]
    save("reference_testdata_cosine_hang", data=hang)
    save("reference_testdata_bad_data", arr_0=bug.astype(np.int32))
    # tests/test_pynndescent_.py:352-369: near-duplicate 32 x 2 points, cosine, k=10, n_trees=20
    r = _class_build(ref, near, "cosine", 10, 189212, 1, n_trees=20)
    save("reference_testdata_cosine_near_duplicates", data=near, seed=np.int64(189212), k=np.int32(10), **r)
    # tests/test_pynndescent_.py:317-349: deduplicated first 1000 rows, cosine, k=10, n_trees=20, recall >= 0.95
    data = np.unique(hang, axis=0)
    data = data[~np.all(data == 0, axis=1)][:1000]
    t0 = time.time()
    r = _class_build(ref, data, "cosine", 10, 189212, 1, n_trees=20)
    print("dedup hang %.1fs" % (time.time() - t0))
    save("class_dedup_hang_cosine", seed=np.int64(189212), k=np.int32(10), **r)


def gen_search_graph(ref):
    """BASELINE config 5's extra pass: the pruning part of NNDescent._init_search_graph (pynndescent_.py:1451-1611),
    executed with the reference's own functions and scipy glue on a reference-built graph; the tree-order
    reordering (1629-1651) is left out."""
    import numba
    import scipy.sparse as sp
    from pynndescent import distances as D
    from pynndescent import pynndescent_ as P

    numba.set_num_threads(1)
    out = {}
    for metric, seed in (("euclidean", 41), ("cosine", 42)):
        x = clustered(1200, 12, 5, 15, seed=seed)
        index = ref.NNDescent(x, metric=metric, n_neighbors=15, random_state=np.random.RandomState(7))
        idx, dst = index._neighbor_graph
        dist = D.fast_distance_alternatives[metric]["dist"]
        rows, dd = P.diversify(idx.copy(), dst.copy(), x, dist, index.rng_state, 1.0)
        fwd_rows, fwd_dist = rows.copy(), dd.copy()
        n = x.shape[0]
        g = sp.coo_array((n, n), dtype=np.float32)
        dd[dd == 0.0] = P.FLOAT32_EPS
        g.row = np.repeat(np.arange(n, dtype=np.int32), rows.shape[1])
        g.col = rows.ravel()
        g.data = dd.ravel()
        g = g.tocsr()
        g.data[g.indices == -1] = 0.0
        g.eliminate_zeros()
        rev = g.transpose()
        P.diversify_csr(rev.indptr, rev.indices, rev.data, x, dist, index.rng_state, 1.0)
        rev.eliminate_zeros()
        rev = rev.tocsr()
        rev.sort_indices()
        g = g.tocsr()
        g.sort_indices()
        u = g.maximum(rev).tocsr()
        u.setdiag(0.0)
        u.eliminate_zeros()
        pre_prune = u.nnz
        u = P.degree_prune(u, int(np.round(1.5 * 15)))
        u.eliminate_zeros()
        u = (u != 0).astype(np.uint8).tocsr()
        u.sort_indices()
        out.update({metric + "_gen": np.array([1200, 12, 5, 15, seed]), metric + "_idx": idx, metric + "_dist": dst,
                    metric + "_fwd_rows": fwd_rows, metric + "_fwd_dist": fwd_dist, metric + "_rev_nnz": np.int64(rev.nnz),
                    metric + "_pre_prune_nnz": np.int64(pre_prune), metric + "_indptr": u.indptr.astype(np.int32),
                    metric + "_indices": u.indices.astype(np.int32)})
    save("search_graph", **out)


def gen_search_graph_modes(ref):
    """The non-default pruning modes of NNDescent._init_search_graph (pynndescent_.py:1451-1611) with the reference's own
    functions and scipy glue, INCLUDING its array aliasing (reverse_graph = self._search_graph.transpose() shares the
    forward matrix's arrays, so the reverse pass prunes the forward matrix too): diversify_prob = 0.5 (standard method;
    coins drawn serially from rng_state) and diversify_method = 'degree_aware' (aggressiveness 2.0)."""
    import numba
    import scipy.sparse as sp
    from pynndescent import distances as D
    from pynndescent import pynndescent_ as P

    numba.set_num_threads(1)
    out = {}
    for tag, metric, seed, method, prob, aggr in (("prob_euclidean", "euclidean", 51, "standard", 0.5, 1.0),
                                                  ("prob_cosine", "cosine", 52, "standard", 0.5, 1.0),
                                                  ("aware_euclidean", "euclidean", 53, "degree_aware", 1.0, 2.0),
                                                  ("aware_cosine", "cosine", 54, "degree_aware", 1.0, 2.0)):
        x = clustered(1200, 12, 5, 15, seed=seed)
        index = ref.NNDescent(x, metric=metric, n_neighbors=15, random_state=np.random.RandomState(7))
        idx, dst = index._neighbor_graph
        dist = D.fast_distance_alternatives[metric]["dist"]
        rng = np.asarray(index.rng_state, np.int64).copy()
        rng_before = rng.copy()
        k = 15
        if method == "degree_aware":
            rows, dd = P.diversify_degree_aware(idx.copy(), dst.copy(), x, dist, int(1.5 * k), aggr, prob)
        else:
            rows, dd = P.diversify(idx.copy(), dst.copy(), x, dist, rng, prob)
        fwd_rows = rows.copy()
        n = x.shape[0]
        g = sp.coo_array((n, n), dtype=np.float32)
        dd[dd == 0.0] = P.FLOAT32_EPS
        g.row = np.repeat(np.arange(n, dtype=np.int32), rows.shape[1])
        g.col = rows.ravel()
        g.data = dd.ravel()
        g = g.tocsr()
        g.data[g.indices == -1] = 0.0
        g.eliminate_zeros()
        fwd_nnz = int(g.nnz)
        rev = g.transpose()
        if method == "degree_aware":
            P.diversify_csr_degree_aware(rev.indptr, rev.indices, rev.data, x, dist, rng, k, aggr, prob)
        else:
            P.diversify_csr(rev.indptr, rev.indices, rev.data, x, dist, rng, prob)
        rev.eliminate_zeros()
        rev_nnz = int(rev.nnz)
        rev = rev.tocsr()
        rev.sort_indices()
        g = g.tocsr()
        g.sort_indices()
        u = g.maximum(rev).tocsr()
        u.setdiag(0.0)
        u.eliminate_zeros()
        pre_prune = u.nnz
        u = P.degree_prune(u, int(np.round(1.5 * k)))
        u.eliminate_zeros()
        u = (u != 0).astype(np.uint8).tocsr()
        u.sort_indices()
        print(tag, "fwd nnz", fwd_nnz, "after reverse pass", rev_nnz, "forward matrix nnz now", int(g.nnz), "final", int(u.nnz))
        out.update({tag + "_gen": np.array([1200, 12, 5, 15, seed]), tag + "_idx": idx, tag + "_dist": dst,
                    tag + "_rng": rng_before, tag + "_fwd_rows": fwd_rows, tag + "_fwd_nnz": np.int64(fwd_nnz),
                    tag + "_rev_nnz": np.int64(rev_nnz), tag + "_pre_prune_nnz": np.int64(pre_prune),
                    tag + "_indptr": u.indptr.astype(np.int32), tag + "_indices": u.indices.astype(np.int32)})
    save("search_graph_modes", **out)


def gen_hub_tree(ref):
    """The graph-informed search tree of NNDescent.prepare(): the reference's own make_hub_tree + convert_tree_format
    (rp_trees.py:1233-1312, 3019-3049) on a reference-built graph, and the whole prepared state of the class
    (vertex order, reordered search graph) for the reordering step (pynndescent_.py:1629-1651)."""
    import numba
    from pynndescent import rp_trees as R

    numba.set_num_threads(1)
    out = {}
    for metric, seed in (("euclidean", 61), ("cosine", 62)):
        x = clustered(2500, 16, 5, 20, seed=seed)
        index = ref.NNDescent(x, metric=metric, n_neighbors=15, random_state=np.random.RandomState(9))
        idx, dst = (np.asarray(a).copy() for a in index._neighbor_graph)
        rng = np.asarray(index.rng_state, np.int64).copy()
        tree = R.make_hub_tree(x, idx, rng.copy(), leaf_size=30, angular=(metric == "cosine"), max_depth=200)
        flat = R.convert_tree_format(tree, x.shape[0], x.shape[1])
        index.prepare()
        sg = index._search_graph.tocsr()
        sg.sort_indices()
        print("hub tree", metric, "nodes", flat.hyperplanes.shape[0], "leaf_size", flat.leaf_size, "search graph nnz", sg.nnz)
        out.update({metric + "_gen": np.array([2500, 16, 5, 20, seed]), metric + "_idx": idx.astype(np.int32),
                    metric + "_dist": dst.astype(np.float32), metric + "_rng": rng,
                    metric + "_hyperplanes": np.asarray(flat.hyperplanes, np.float32),
                    metric + "_offsets": np.asarray(flat.offsets, np.float32),
                    metric + "_children": np.asarray(flat.children, np.int32),
                    metric + "_indices": np.asarray(flat.indices, np.int32), metric + "_leaf_size": np.int32(flat.leaf_size),
                    metric + "_vertex_order": np.asarray(index._vertex_order, np.int32),
                    metric + "_sg_indptr": sg.indptr.astype(np.int32), metric + "_sg_indices": sg.indices.astype(np.int32),
                    metric + "_prepared_tree_indices": np.asarray(index._search_forest[0].indices, np.int32),
                    metric + "_raw_after": np.asarray(index._raw_data, np.float32)})
        # the reference's own queries on the prepared index (query path: pynndescent_.py:2275-2379)
        q = clustered(200, 16, 5, 20, seed=seed + 100)
        qi, qd = index.query(q, k=10, epsilon=0.1)
        out.update({metric + "_queries": q, metric + "_query_idx": qi.astype(np.int32), metric + "_query_dist": qd.astype(np.float32)})
    save("hub_tree", **out)


def gen_update(ref):
    """NNDescent.update (pynndescent_.py:2381-2553): fresh rows appended + some rows replaced, warm start from the
    old graph (flag 0) + a smaller forest, no random init."""
    import numba

    for metric, seed in (("euclidean", 11), ("cosine", 12)):
        x = clustered(1500, 12, 5, 25, seed=31)
        rs = np.random.RandomState(1000 + seed)
        fresh = (x[rs.choice(1500, 200, replace=False)] + 0.05 * rs.standard_normal((200, 12))).astype(np.float32)
        upd_idx = np.sort(rs.choice(1500, 30, replace=False)).astype(np.int64)
        upd = (x[rs.choice(1500, 30, replace=False)] + 0.05 * rs.standard_normal((30, 12))).astype(np.float32)
        numba.set_num_threads(2)
        t0 = time.time()
        index = ref.NNDescent(x.copy(), metric=metric, n_neighbors=10, n_trees=6, random_state=np.random.RandomState(seed),
                              n_iters=8)
        before = (np.asarray(index._neighbor_graph[0], np.int32).copy(), np.asarray(index._neighbor_graph[1], np.float32).copy())
        rng_after_build = np.asarray(index.rng_state, np.int64).copy()
        index.update(xs_fresh=fresh, xs_updated=upd, updated_indices=upd_idx)
        after = (np.asarray(index._neighbor_graph[0], np.int32), np.asarray(index._neighbor_graph[1], np.float32))
        numba.set_num_threads(1)
        print("update", metric, "%.1fs" % (time.time() - t0), "n after", after[0].shape)
        save("update_%s_T2" % metric, gen=np.array([1500, 12, 5, 25, 31]), seed=np.int64(seed), k=np.int32(10),
             n_trees=np.int32(6), n_iters=np.int32(8), n_threads=np.int32(2), fresh=fresh, upd=upd, upd_idx=upd_idx,
             before_idx=before[0], before_dist=before[1], after_idx=after[0], after_dist=after[1],
             rng_after_build=rng_after_build, raw_after=np.asarray(index._raw_data, np.float32))


GENERATORS = {
    "primitives": gen_primitives,
    "rp": gen_rp,
    "candidates": gen_candidates,
    "build_small": gen_build_small,
    "build_clustered": gen_build_clustered,
    "build_c1": gen_build_c1,
    "build_wide": gen_build_wide,
    "reference_testdata": gen_reference_testdata,
    "search_graph": gen_search_graph,
    "search_graph_modes": gen_search_graph_modes,
    "hub_tree": gen_hub_tree,
    "update": gen_update,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    ref = ref_t0.load_reference()
    for name, fn in GENERATORS.items():
        if args.only and name not in args.only:
            continue
        t0 = time.time()
        fn(ref)
        print("[%s] %.1fs" % (name, time.time() - t0))


if __name__ == "__main__":
    main()

"""Pins the CPU oracle (oracle/nnd_oracle.c) against fixtures produced by the
reference itself (tests/golden/make_golden.py, reference source run un-jitted).

Bit-exact where the arithmetic is integer / sequential f32; tolerances are stated
where libm or accumulation width legitimately differs."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O


def _load(golden_dir, name):
    path = os.path.join(golden_dir, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("fixture %s missing" % name)
    return np.load(path)


def test_tau_rand_bit_exact(golden_dir, oracle_strict):
    g = _load(golden_dir, "primitives")
    lib = oracle_strict
    for r in range(g["states0"].shape[0]):
        s = g["states0"][r].copy()
        got = np.array([lib.orc_tau_rand_int(s) for _ in range(40)], np.int32)
        np.testing.assert_array_equal(got, g["ints"][r])
        np.testing.assert_array_equal(s, g["states_after"][r])
        s = g["states0"][r].copy()
        gotf = np.array([lib.orc_tau_rand(s) for _ in range(40)], np.float32)
        np.testing.assert_array_equal(gotf, g["floats"][r])


def test_distances(golden_dir, oracle_strict):
    g = _load(golden_dir, "primitives")
    lib = oracle_strict
    xs, ys = g["xs"], g["ys"]
    d = xs.shape[1]
    sq = np.array([lib.orc_squared_euclidean(x, y, d) for x, y in zip(xs, ys)], np.float32)
    np.testing.assert_array_equal(sq, g["sq"])  # sequential f32 accumulation: bit-exact
    ac = np.array([lib.orc_alternative_cosine(x, y, d) for x, y in zip(xs, ys)], np.float32)
    # zero-vector / non-positive-dot conventions must agree exactly (FLT_MAX, 0)
    big = g["ac"] > 1e30
    np.testing.assert_array_equal(ac > 1e30, big)
    np.testing.assert_array_equal(ac[g["ac"] == 0], 0.0)
    # finite values: libm log2f vs numpy's float32 log2 loop may differ by an ulp
    np.testing.assert_allclose(ac[~big], g["ac"][~big], rtol=3e-7, atol=3e-7)
    eu = np.array([lib.orc_euclidean(x, y, d) for x, y in zip(xs, ys)])
    co = np.array([lib.orc_cosine(x, y, d) for x, y in zip(xs, ys)])
    np.testing.assert_allclose(eu, g["eu"], rtol=1e-5)  # reference accumulates in f32 here
    np.testing.assert_allclose(co, g["co"], rtol=1e-5, atol=2e-6)
    # correction(alt) ~= true distance (reference tests/test_distances.py:326-343)
    np.testing.assert_allclose(O.correct_distances(g["sq"], "euclidean"), g["corr_e"], rtol=1e-6)
    m = g["corr_c_mask"]
    np.testing.assert_allclose(O.correct_distances(np.where(m, g["ac"], 0), "cosine")[m], g["corr_c"][m],
                               rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(O.correct_distances(sq, "euclidean"), eu, rtol=2e-5)
    np.testing.assert_allclose(O.correct_distances(ac, "cosine")[~big], co[~big], rtol=1e-4, atol=1e-6)


def test_heap_push_bit_exact(golden_dir, oracle_strict):
    g = _load(golden_dir, "primitives")
    lib = oracle_strict
    size = g["snap_i"].shape[1]
    hi = np.full(size, -1, np.int32)
    hd = np.full(size, np.inf, np.float32)
    hf = np.zeros(size, np.uint8)
    for t in range(g["push_p"].shape[0]):
        r = lib.orc_checked_flagged_heap_push(hd, hi, hf, size, float(g["push_p"][t]), int(g["push_n"][t]),
                                              int(g["push_f"][t]))
        assert r == g["push_ret"][t]
        np.testing.assert_array_equal(hi, g["snap_i"][t])
        np.testing.assert_array_equal(hd, g["snap_d"][t])
        np.testing.assert_array_equal(hf, g["snap_f"][t])
    pri = np.full(size, np.inf, np.float32)
    ind = np.full(size, -1, np.int32)
    for t in range(g["push_p"].shape[0]):
        r = lib.orc_checked_heap_push(pri, ind, size, float(g["push_p"][t]), int(g["push_n"][t]))
        assert r == g["push_ret_unflagged"][t]
    np.testing.assert_array_equal(pri, g["unflagged_pri"])
    np.testing.assert_array_equal(ind, g["unflagged_ind"])


def test_deheap_sort_bit_exact(golden_dir, oracle_strict):
    g = _load(golden_dir, "primitives")
    i = g["sort_pre_i"].copy()
    d = g["sort_pre_d"].copy()
    oracle_strict.orc_deheap_sort(i, d, i.shape[0], i.shape[1])
    np.testing.assert_array_equal(i, g["sort_i"])
    np.testing.assert_array_equal(d, g["sort_d"])
    # rows ascending, empty slots (-1, inf) last
    np.testing.assert_array_equal(d, np.sort(d, axis=1))


def _split(lib, fn, data, indices, state):
    m = indices.shape[0]
    side = np.zeros(m, np.int8)
    hyper = np.zeros(data.shape[1], np.float32)
    off = C.c_float()
    st = state.copy()
    nl = fn(np.ascontiguousarray(data), data.shape[1], indices, m, st, side, hyper, C.byref(off))
    return indices[side == 0], indices[side == 1], hyper, off.value, st, nl


def test_euclidean_split_bit_exact(golden_dir, oracle_strict):
    g = _load(golden_dir, "rp")
    for trial in range(3):
        left, right, hyper, off, st, nl = _split(oracle_strict, oracle_strict.orc_euclidean_split, g["data"],
                                                 g["indices"], g["euclid_%d_state_in" % trial])
        np.testing.assert_array_equal(left, g["euclid_%d_left" % trial])
        np.testing.assert_array_equal(right, g["euclid_%d_right" % trial])
        np.testing.assert_array_equal(hyper, g["euclid_%d_hyper" % trial])
        np.testing.assert_array_equal(st, g["euclid_%d_state_out" % trial])
        assert nl == left.shape[0]
        np.testing.assert_allclose(off, float(g["euclid_%d_off" % trial]), rtol=1e-6)


def test_angular_split(golden_dir, oracle_strict):
    # numba accumulates `norm` in f64 (utils.py:70 comments the f32 local out) while the
    # un-jitted run accumulates in f32, so the hyperplane may differ in the last ulps:
    # require identical partitions up to points whose margin is ~0.
    g = _load(golden_dir, "rp")
    for trial in range(3):
        left, right, hyper, off, st, nl = _split(oracle_strict, oracle_strict.orc_angular_split, g["data"],
                                                 g["indices"], g["angular_%d_state_in" % trial])
        np.testing.assert_allclose(hyper, g["angular_%d_hyper" % trial], rtol=0, atol=2e-6)
        ref_left = set(g["angular_%d_left" % trial].tolist())
        mism = len(set(left.tolist()) ^ ref_left)
        assert mism <= 6, mism  # the zero rows (coin flips after differing RNG use) at most
        assert off == 0.0


def test_degenerate_split(golden_dir, oracle_strict):
    g = _load(golden_dir, "rp")
    data = g["degenerate_data"]
    ind = np.arange(50, dtype=np.int32)
    left, right, *_ = _split(oracle_strict, oracle_strict.orc_euclidean_split, data, ind,
                             np.array([12345, 67890, 13579], np.int64))
    np.testing.assert_array_equal(left, g["degenerate_left"])
    np.testing.assert_array_equal(right, g["degenerate_right"])


def test_forest_leaf_array(golden_dir, oracle_strict):
    g = _load(golden_dir, "rp")
    fdata = g["forest_data"]
    # make_forest draws the per-tree states from the RandomState it is given (rp_trees.py:2850)
    states = np.random.RandomState(5).randint(O.INT32_MIN, O.INT32_MAX, size=(3, 3)).astype(np.int64)
    la = O.make_leaf_array(fdata, 3, 25, states, angular=False, lib=oracle_strict)
    np.testing.assert_array_equal(la, g["forest_euclid_leaf_array"])  # bit-exact
    la_a = O.make_leaf_array(fdata, 3, 25, states, angular=True, lib=oracle_strict)
    ref = g["forest_angular_leaf_array"]
    assert la_a.shape[1] == ref.shape[1]
    assert abs(la_a.shape[0] - ref.shape[0]) <= max(3, ref.shape[0] // 10)
    for arr in (la, la_a):  # every tree partitions the point set
        ids = arr[arr >= 0]
        assert ids.shape[0] == 3 * fdata.shape[0]
        assert np.all(np.bincount(ids, minlength=fdata.shape[0]) == 3)
    states = np.random.RandomState(6).randint(O.INT32_MIN, O.INT32_MAX, size=(2, 3)).astype(np.int64)
    la3 = O.make_leaf_array(fdata, 2, 25, states, angular=False, max_depth=3, lib=oracle_strict)
    np.testing.assert_array_equal(la3, g["forest_depth3_leaf_array"])
    assert la3.shape[1] > 25  # depth-limited leaves exceed leaf_size (rp_trees.py:2548-2551)
    assert O.make_leaf_array(fdata, 0, 25, np.zeros((0, 3), np.int64), False).tolist() == [[-1]]


@pytest.mark.parametrize("T", [1, 3])
def test_new_build_candidates_bit_exact(golden_dir, oracle_strict, T):
    g = _load(golden_dir, "candidates")
    hi = g["heap_i"].copy()
    hf = g["heap_f"].copy()
    n, k = hi.shape
    mc = int(g["mc"])
    new = np.empty((n, mc), np.int32)
    old = np.empty((n, mc), np.int32)
    oracle_strict.orc_new_build_candidates(hi, hf, n, k, mc, g["rng_state"].copy(), T, new, old)
    np.testing.assert_array_equal(new, g["T%d_new" % T])
    np.testing.assert_array_equal(old, g["T%d_old" % T])
    np.testing.assert_array_equal(hf, g["T%d_flags_after" % T])

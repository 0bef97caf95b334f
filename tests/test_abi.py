"""The C-ABI library loads and exports every symbol include/pynnd_amd.h declares (no compute, no GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pynnd_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nnd_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from pynndescent_amd import _capi

    lib = _capi.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "libpynnd_amd.so does not export %s" % name
    assert sorted(_capi.EXPORTED_SYMBOLS) == declared
    assert lib.nnd_abi_version() == 6


def test_no_cpu_fallback_without_a_device():
    """Without a gfx950 device the product path fails loudly (it must never route through the oracle)."""
    import torch

    from pynndescent_amd import _capi

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.NNDError, match="no HIP device|gfx950"):
        _capi.Builder(100, 8, 0, 10, 2, 60, 200, 10, 5, 0.001, [1, 2, 3], [4, 5, 6])


def test_multi_gpu_entry_point_fails_loudly_without_a_device():
    """nnd_build_multi (the n_devices form of the drop-in call) has no CPU path either."""
    import ctypes as C

    import numpy as np
    import torch

    from pynndescent_amd import _capi

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _capi.load_library()
    p = _capi.NNDParams()
    p.n, p.dim, p.metric, p.n_neighbors, p.n_trees, p.leaf_size = 64, 4, 0, 5, 1, 60
    p.max_depth, p.max_candidates, p.n_iters, p.delta = 200, 5, 3, 0.001
    x = np.zeros((64, 4), np.float32)
    idx = np.empty((64, 5), np.int32)
    dist = np.empty((64, 5), np.float32)
    err = C.create_string_buffer(512)
    rc = lib.nnd_build_multi(C.byref(p), x.ctypes.data_as(C.c_void_p), 2, None, idx.ctypes.data_as(C.c_void_p),
                             dist.ctypes.data_as(C.c_void_p), None, None, err, 512)
    assert rc != 0 and b"no HIP device" in err.value


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pynndescent_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "nnd_oracle" not in src, f


def test_constructor_mirrors_reference_signature():
    """Positional order of the reference ctor (pynndescent_.py:976-1007): the tests' positional 10 is bit_metric."""
    import inspect

    from pynndescent_amd import NNDescent

    names = list(inspect.signature(NNDescent.__init__).parameters)[1:]
    assert names[:12] == ["data", "metric", "metric_kwds", "bit_metric", "n_neighbors", "n_trees", "angular_trees",
                          "leaf_size", "pruning_degree_multiplier", "diversify_prob", "diversify_method",
                          "degree_prune_aggressiveness"]
    for kw in ("tree_init", "init_graph", "init_dist", "random_state", "low_memory", "max_candidates", "max_rptree_depth",
               "n_iters", "delta", "n_jobs", "compressed", "parallel_batch_queries", "verbose"):
        assert kw in names
    sig = inspect.signature(NNDescent.__init__).parameters
    assert sig["n_neighbors"].default == 30 and sig["delta"].default == 0.001 and sig["max_rptree_depth"].default == 200


def test_from_graph_validates_and_mirrors_reference_attributes():
    """NNDescent.from_graph needs no GPU: it wraps an existing graph; shapes / metric are validated like the ctor's
    init_graph branch (pynndescent_.py:1229, 1292) and the attributes the reference's prepare()/query() read exist."""
    import numpy as np
    import pytest

    from pynndescent_amd import NNDescent

    x = np.random.RandomState(0).standard_normal((50, 4)).astype(np.float32)
    idx = np.tile(np.arange(5, dtype=np.int32), (50, 1))
    dist = np.zeros((50, 5), np.float32)
    index = NNDescent.from_graph(x, idx, dist, metric="cosine", random_state=3, leaf_size=20)
    assert index.n_neighbors == 5 and index._angular_trees and index.leaf_size == 20
    assert index.rng_state.shape == (3,) and index.search_rng_state.shape == (3,)
    for name in NNDescent._HANDOVER:
        assert hasattr(index, name), name
    with pytest.raises(ValueError, match="Init graph size does not match"):
        NNDescent.from_graph(x, idx[:10], dist[:10])
    with pytest.raises(ValueError, match="Metric is neither callable"):
        NNDescent.from_graph(x, idx, dist, metric="no-such-metric")
    with pytest.raises(TypeError):
        NNDescent.from_graph(x, idx, dist, not_a_parameter=1)


def test_host_copy_and_sqrt_helpers_match_numpy():
    """nnd_host_copy / nnd_host_sqrt_f32 (the threaded first-touch helpers behind NNDescent.neighbor_graph): the same
    bytes as ndarray.copy() and numpy.sqrt on float32; small or non-contiguous inputs take numpy's own path."""
    import numpy as np

    from pynndescent_amd import _capi

    rs = np.random.RandomState(3)
    a = rs.randint(-1, 1 << 30, size=(700_001, 3)).astype(np.int32)  # > 4 MB, not a multiple of the page size
    b = _capi.host_copy(a)
    assert b is not a and b.dtype == a.dtype and np.array_equal(a, b)
    d = np.abs(rs.standard_normal((1_100_003,)).astype(np.float32)) * 1e3
    d[:5] = [0.0, np.inf, 1e-45, 3.4e38, 2.0]
    s = _capi.host_sqrt(d)
    assert s.dtype == np.float32 and np.array_equal(s.view(np.uint32), np.sqrt(d).view(np.uint32))
    small = np.arange(10, dtype=np.float32)
    assert np.array_equal(_capi.host_sqrt(small), np.sqrt(small))
    assert np.array_equal(_capi.host_copy(a[::2]), a[::2])


def test_host_pool_falls_back_to_numpy_without_a_device():
    """_capi.HostPool (round 6): result arrays in pinned host memory -- without a HIP device nnd_host_alloc returns NULL and the
    pool hands out ordinary numpy arrays; small requests never touch the library."""
    import numpy as np
    import torch

    from pynndescent_amd import _capi

    pool = _capi.HostPool()
    small = pool.empty((10, 3), np.float32)
    assert small.shape == (10, 3) and small.flags.owndata
    big = pool.empty((1_200_000, 2), np.int32)  # 9.6 MB
    assert big.shape == (1_200_000, 2) and big.dtype == np.int32 and big.flags.writeable and big.flags.c_contiguous
    big[:] = 7
    assert int(big.sum()) == 7 * big.size
    if not torch.cuda.is_available():
        assert big.flags.owndata  # numpy's own memory: nothing to return to the pool
        lib = _capi.load_library()
        assert not lib.nnd_host_alloc(1 << 20)
    pool.trim()

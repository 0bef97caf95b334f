"""Helpers for the -m gpu tests (they call the HIP path through the C ABI only)."""
import numpy as np

from oracle import oracle as O
from pynndescent_amd import _capi


def alt_dist_matrix(x, rows_a, rows_b, metric):
    """float64 alt-space distances (reference distances.py:63-91, 583-630)."""
    a = x[rows_a].astype(np.float64)
    b = x[rows_b].astype(np.float64)
    if metric == "euclidean":
        return ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    dot = a @ b.T
    na = (a * a).sum(1)[:, None]
    nb = (b * b).sum(1)[None, :]
    out = np.full(dot.shape, 3.402823466e38)
    ok = (na > 0) & (nb > 0) & (dot > 0)
    with np.errstate(divide="ignore", invalid="ignore"):
        out[ok] = np.log2(np.sqrt(na * nb)[ok] / dot[ok])
    out[(na == 0) & (nb == 0)] = 0.0
    return np.maximum(out, 0.0)


def make_builder(x, metric="euclidean", k=15, n_trees=8, leaf_size=None, mc=None, n_iters=None, delta=0.001,
                 seed=1, max_depth=200, join_blocks=1, flags=0):
    n, d = x.shape
    rng_state, _, tree_states = O.draw_rng_states(seed, max(n_trees, 1))
    ls = O.default_leaf_size(k) if leaf_size is None else leaf_size
    mc = min(60, k) if mc is None else mc
    n_iters = O.default_n_iters(n) if n_iters is None else n_iters
    b = _capi.Builder(n, d, O.METRICS[metric], k, n_trees, ls, max_depth, mc, n_iters, delta, rng_state,
                      tree_states[0], join_blocks=join_blocks, flags=flags)
    b.set_data_host(x)
    return b


def check_graph_invariants(x, metric, idx, dist, tol=2e-4, atol=1e-5, name=""):
    """rows ascending, ids unique, stored alt distances match the true ones for the stored ids."""
    n, k = idx.shape
    d64 = np.where(np.isfinite(dist), dist.astype(np.float64), 1e39)
    assert np.all(np.diff(d64, axis=1) >= 0), name + ": rows not ascending"
    for r in range(n):
        v = idx[r][idx[r] >= 0]
        assert len(v) == len(np.unique(v)), "%s: duplicate ids in row %d: %s" % (name, r, idx[r])
    rows = np.arange(n)
    valid = idx >= 0
    xi = x.astype(np.float64)
    if metric == "euclidean":
        true = ((xi[:, None, :] - xi[np.where(valid, idx, 0)]) ** 2).sum(-1)
    else:
        nb = xi[np.where(valid, idx, 0)]
        dot = (xi[:, None, :] * nb).sum(-1)
        na = (xi * xi).sum(1)[:, None]
        nbn = (nb * nb).sum(-1)
        true = np.full(dot.shape, 3.402823466e38)
        ok = (na > 0) & (nbn > 0) & (dot > 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            true[ok] = np.log2(np.sqrt(na * nbn)[ok] / dot[ok])
        true[(na == 0) & (nbn == 0)] = 0.0
        true = np.maximum(true, 0)
    big = true > 1e30
    m = valid & ~big
    scale = max(1.0, float(np.abs(true[m]).max())) if m.any() else 1.0
    err = np.abs(dist[m].astype(np.float64) - true[m])
    lim = tol * np.abs(true[m]) + atol * scale
    assert (err <= lim).all(), "%s: stored distances off, max err %g (rel %g)" % (
        name, err.max(), (err / np.maximum(np.abs(true[m]), 1e-30)).max())
    assert np.all(dist[valid & big] > 1e30), name + ": FLT_MAX convention"
    assert np.all(np.isinf(dist[~valid])), name + ": empty slots must be +inf"


# ---- full-size configurations: ONE generator and ONE oracle run per configuration and test SESSION, shared by the test
# ---- modules (tests/test_gpu_fullsize.py, tests/test_gpu_sharded.py): the 10 M-point oracle build takes minutes
def gen_fullsize(n, d, latent, seed, dev, nonneg):
    """The full-size synthetic sets (SURVEY.md section 8d generator), generated ON the device: deterministic for a seed."""
    import torch

    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    centres = torch.randn(1024, latent, generator=g, device=dev) * 3.0
    proj = torch.randn(latent, d, generator=g, device=dev) / latent ** 0.5
    assign = torch.randint(0, 1024, (n,), generator=g, device=dev)
    x = (centres[assign] + torch.randn(n, latent, generator=g, device=dev)) @ proj
    x = x + 0.3 * torch.randn(n, d, generator=g, device=dev)
    if nonneg:
        x = (x + 12.0).clamp_min(0) * 9.0
    return x.contiguous()


# name -> (n, d, latent, generator seed, non-negative, metric, k, n_trees, oracle threads)
FULLSIZE = {
    "c2": (1_000_000, 128, 16, 1, True, "euclidean", 15, 8, 64),
    "c3": (1_200_000, 100, 24, 2, False, "cosine", 15, 12, 64),
    "c5": (290_000, 256, 32, 4, False, "cosine", 15, 11, 64),
    "3m": (3_000_000, 128, 16, 3, True, "euclidean", 15, 12, 128),
    "c4": (10_000_000, 128, 16, 3, True, "euclidean", 15, 12, 128),
    # round 6: the other workloads of the bench line -- the slow-converging set (latent dimension 48) and the reference's default k = 30
    "hard": (1_000_000, 128, 48, 1, True, "euclidean", 15, 8, 64),
    "k30": (1_000_000, 128, 16, 1, True, "euclidean", 30, 8, 64),
}
_ORACLE_CACHE = {}


def fullsize_points(name, dev):
    n, d, latent, seed, nonneg = FULLSIZE[name][:5]
    return gen_fullsize(n, d, latent, seed, dev, nonneg)


def fullsize_oracle(name, x_host=None, dev=None):
    """(indices, distances, seconds) of the CPU oracle (the reference algorithm, random_state 1) on configuration `name`; computed
    once per session -- by whichever test asks first -- and kept."""
    import time

    if name not in _ORACLE_CACHE:
        _, _, _, _, _, metric, k, n_trees, n_threads = FULLSIZE[name]
        if x_host is None:
            x_host = fullsize_points(name, dev).cpu().numpy()
        t0 = time.perf_counter()
        oi, od = O.build_index(x_host, metric, n_neighbors=k, n_trees=n_trees, random_state=1, n_threads=n_threads, kind="fast")
        _ORACLE_CACHE[name] = (oi, od, time.perf_counter() - t0)
    return _ORACLE_CACHE[name]


def two_sided(x_host, metric, gpu_idx, oracle_idx, n_rows=1000, band=0.005, seed=5):
    """recall@10 of both sides on the same sampled rows against exact brute force: |GPU - reference algorithm| <= band."""
    rows = np.random.RandomState(seed).choice(x_host.shape[0], n_rows, replace=False)
    ti, _ = O.brute_force_knn(x_host, 10, metric, rows=rows, kind="fast")
    r_gpu, r_cpu = O.recall(ti, gpu_idx[rows]), O.recall(ti, oracle_idx[rows])
    assert abs(r_gpu - r_cpu) <= band, (r_gpu, r_cpu)  # north star: within +-0.5 % of the reference algorithm
    return r_gpu, r_cpu

"""Seeded synthetic inputs shared by tests, bench.py and the golden generator
(tests/golden/make_golden.py holds identical copies of clustered / nn_data_like)."""
import numpy as np


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
    """Shape/convention of the reference fixture nn_data (reference tests/conftest.py:47-52)."""
    rs = np.random.RandomState(seed)
    x = rs.uniform(0, 1, size=(1000, 5))
    return np.vstack([x, np.zeros((2, 5))]).astype(np.float32)

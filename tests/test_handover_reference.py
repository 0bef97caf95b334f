"""to_reference(): a graph built here continues in the reference class without a rebuild (prepare + query).
Runs only where the reference is importable (the build container: /root/reference through the T0 stub); it needs no
GPU -- the index object is made around a graph fixture with NNDescent.from_graph."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util_data import nn_data_like

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _reference():
    try:
        from oracle import ref_t0

        return ref_t0.load_reference()
    except Exception as e:  # pragma: no cover
        pytest.skip("reference not importable here: %s" % e)


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_handover_prepare_and_query(metric):
    ref = _reference()
    from pynndescent_amd import NNDescent

    x = nn_data_like()
    g = np.load(os.path.join(GOLDEN, "class_nndata_%s.npz" % metric))
    index = NNDescent.from_graph(x, g["idx"], g["dist"], metric=metric, random_state=7)
    handed = index.to_reference()
    assert type(handed) is ref.NNDescent
    q = x[:12] + 0.01
    idx, dist = handed.query(q, k=5, epsilon=0.2)  # runs the reference's prepare(): hub tree, search graph, search
    assert idx.shape == (12, 5)
    ti, td = O.brute_force_knn(np.vstack([x, q]), 6, metric, rows=np.arange(len(x), len(x) + 12))
    truth = [[j for j in row if j < len(x)][:5] for row in ti]
    rec = np.mean([len(np.intersect1d(t, a)) / 5.0 for t, a in zip(truth, idx)])
    assert rec >= 0.9, rec
    assert np.all(np.diff(dist, axis=1) >= 0)
    cg = handed.neighbor_graph  # the reference's own accessor on the handed-over graph
    assert cg[0].shape == g["idx"].shape

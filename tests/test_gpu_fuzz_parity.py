"""Randomised parity sweep (tools/fuzz_parity.py): GPU build vs the oracle's NN-descent on the GPU's own leaf array over
random (n, d, k, metric, n_trees, leaf_size, max_candidates) -- 12 configurations per run, recall within 0.03."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_configurations():
    import fuzz_parity

    assert fuzz_parity.main(12, 7) == 0


def test_structural_stress():
    """tools/fuzz_structural.py: 60 random configurations (duplicates, d up to 513, k up to 64, leaf sizes 2..300,
    0..16 trees, 1..64 candidates): shapes, ascending rows, unique ids, finite distances, no exception."""
    import fuzz_structural

    assert fuzz_structural.main(60, 11) == 0


def test_structural_stress_wide_rows_and_candidate_lists():
    """The same sweep over rows of 65..256 neighbours and candidate lists of 65..128 (round 5's bounds): 16 configurations."""
    import fuzz_structural

    assert fuzz_structural.main(16, 5, wide=True) == 0

"""pynndescent_amd -- MI355X (gfx950) native NN-Descent index builder.

Drop-in for the build path of lmcinnes/pynndescent: ``NNDescent(data, ...).neighbor_graph``.
Host code is Python (like the reference); the hot path is hand-written HIP behind the C ABI in
``include/pynnd_amd.h`` (``libpynnd_amd.so``, bound with ctypes in ``_capi.py``).
"""
from .nndescent import EMPTY_GRAPH, NNDescent, make_index, nn_descent  # noqa: F401

__version__ = "0.1.0"

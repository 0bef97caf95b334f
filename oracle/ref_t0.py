"""Loader for "T0": the reference source run un-jitted -- TEST INFRASTRUCTURE ONLY.

Imports the *unmodified* reference package from ``/root/reference`` (read-only,
build container only -- it does not exist on the GPU box) on top of the stub
``numba`` in ``oracle/numba_stub``.  Used by ``tests/golden/make_golden.py`` to
generate the committed golden fixtures and by CPU tests that are skipped when
``/root/reference`` is absent.  Never imported by ``pynndescent_amd``.
"""
import importlib
import importlib.metadata
import os
import sys

REFERENCE_ROOT = os.environ.get("PYNND_REFERENCE_ROOT", "/root/reference")
_STUB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "numba_stub")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pynndescent"))


def load_reference(n_threads=1):
    """Return the imported reference ``pynndescent`` module (un-jitted)."""
    if not reference_available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # /root/reference is read-only
    if _STUB not in sys.path:
        sys.path.insert(0, _STUB)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    _orig_version = importlib.metadata.version

    def _version(name):  # reference __init__.py:22 asks for its installed version
        if name == "pynndescent":
            return "0.6.0"
        return _orig_version(name)

    importlib.metadata.version = _version
    try:
        mod = importlib.import_module("pynndescent")
    finally:
        importlib.metadata.version = _orig_version
    import numba

    assert getattr(numba, "__version__", "") == "0.0-stub", "real numba shadowed the stub"
    numba.set_num_threads(n_threads)
    return mod

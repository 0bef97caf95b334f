"""Minimal stand-in for ``numba`` -- TEST INFRASTRUCTURE ONLY.

numba/llvmlite are not installable in the build container (no network), so the
reference (``/root/reference/pynndescent``, pure Python + ``@numba.njit``) cannot
be run compiled.  This stub lets the *unmodified* reference source be imported
and executed by the CPython interpreter ("T0" in SURVEY.md section 8c): every
``njit``/``jit`` decorator returns the plain Python function, ``prange`` is
``range``, and the type objects are inert except that

* a string signature with an ``i4`` / ``f4`` return type (``"i4(i8[:])"`` for
  ``tau_rand_int``, ``"f4(i8[:])"`` for ``tau_rand`` -- reference
  ``pynndescent/utils.py:17,43``) is honoured by casting the return value, so
  the Tausworthe generator wraps to int32 exactly as the compiled code does;
* ``get_num_threads()`` returns a settable value so the reference's
  thread-block ownership logic (``utils.py:259-306``, ``utils.py:706-731``)
  can be replayed deterministically for any thread count.

Nothing under ``oracle/`` is imported by the product path.
"""
import numpy as np

__version__ = "0.0-stub"

_NUM_THREADS = [1]


def get_num_threads():
    return _NUM_THREADS[0]


def set_num_threads(n):
    _NUM_THREADS[0] = int(n)


prange = range


class _Config:
    THREADING_LAYER = "workqueue"
    DISABLE_JIT = 1


config = _Config()


def _wrap_cast(x, np_type):
    """Cast with C-style wraparound for integer targets."""
    if np.issubdtype(np_type, np.integer):
        if isinstance(x, (float, np.floating)):
            x = int(x)
        if isinstance(x, np.ndarray):
            return x.astype(np_type)
        info = np.iinfo(np_type)
        span = int(info.max) - int(info.min) + 1
        v = (int(x) - int(info.min)) % span + int(info.min)
        return np_type(v)
    return np_type(x)


class _Sig:
    def __init__(self, ret, args):
        self.ret = ret
        self.args = args


class _Type:
    """Inert numba type: subscriptable, callable as cast or signature builder."""

    def __init__(self, name, np_type=None):
        self.name = name
        self.np_type = np_type

    def __getitem__(self, item):
        return _Type(self.name + "[]", None)

    def __call__(self, *args, **kwargs):
        if len(args) >= 1 and all(isinstance(a, (_Type, _Sig)) for a in args):
            return _Sig(self, args)
        if len(args) == 1 and self.np_type is not None:
            return _wrap_cast(args[0], self.np_type)
        return _Sig(self, args)

    def __repr__(self):
        return "<stub numba type %s>" % self.name


_SCALARS = {
    "boolean": np.bool_,
    "int8": np.int8,
    "int16": np.int16,
    "int32": np.int32,
    "int64": np.int64,
    "intp": np.intp,
    "uint8": np.uint8,
    "uint16": np.uint16,
    "uint32": np.uint32,
    "uint64": np.uint64,
    "uintp": np.uintp,
    "float32": np.float32,
    "float64": np.float64,
}


class _Types:
    def __init__(self):
        for k, v in _SCALARS.items():
            setattr(self, k, _Type(k, v))

    def Array(self, dtype, ndim, layout, readonly=False):
        return _Type("Array")

    def Tuple(self, members):
        return _Type("Tuple")

    def UniTuple(self, t, n):
        return _Type("UniTuple")

    def ListType(self, t):
        return _Type("ListType")

    def List(self, t):
        return _Type("List")

    def __getattr__(self, item):  # anything else: an inert type
        return _Type(item)


types = _Types()
for _k, _v in _SCALARS.items():
    globals()[_k] = getattr(types, _k)


def typeof(x):
    return _Type("typeof")


_RET_CASTS = {"i4": np.int32, "f4": np.float32, "i8": np.int64, "f8": np.float64}


def _ret_cast_from_sig(sig):
    if isinstance(sig, (list, tuple)) and sig:
        sig = sig[0]
    if isinstance(sig, str):
        head = sig.split("(", 1)[0].strip()
        return _RET_CASTS.get(head)
    return None


def _make_decorator(*dargs, **dkwargs):
    # used as @njit, @njit(), @njit("sig", ...), @njit([sigs], ...), @njit(sigobj, ...)
    if len(dargs) == 1 and callable(dargs[0]) and not isinstance(dargs[0], (_Type, _Sig)) and not dkwargs:
        fn = dargs[0]
        fn.py_func = fn
        return fn
    cast = _ret_cast_from_sig(dargs[0]) if dargs else None

    def deco(fn):
        if cast is None:
            fn.py_func = fn
            return fn
        import functools

        @functools.wraps(fn)
        def wrapper(*a, **k):
            return _wrap_cast(fn(*a, **k), cast)

        wrapper.py_func = fn
        return wrapper

    return deco


njit = _make_decorator
jit = _make_decorator
generated_jit = _make_decorator


def vectorize(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not isinstance(dargs[0], (list, str)):
        return np.vectorize(dargs[0])

    def deco(fn):
        return np.vectorize(fn)

    return deco


class _TypedList:
    @staticmethod
    def empty_list(item_type=None, allocated=0):
        return []

    def __call__(self, *a):
        return list(*a)


class _Typed:
    List = _TypedList()

    class Dict:
        @staticmethod
        def empty(k, v):
            return {}


typed = _Typed()

from . import extending  # noqa: E402,F401
from . import core  # noqa: E402,F401

"""Stub of numba.extending (test infrastructure; see numba/__init__.py)."""


def intrinsic(fn):
    # reference distances.py:31-47 defines popcnt_u8 through an LLVM intrinsic;
    # un-jitted we substitute a Python popcount (bit metrics are out of scope).
    def popcount(v):
        return bin(int(v) & 0xFF).count("1")

    return popcount


def overload(*a, **k):
    def deco(fn):
        return fn

    return deco


register_jitable = lambda fn: fn  # noqa: E731

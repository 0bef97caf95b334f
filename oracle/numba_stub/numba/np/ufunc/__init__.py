"""Stub of numba.np.ufunc (test infrastructure): no tbbpool, so the reference falls back to workqueue."""

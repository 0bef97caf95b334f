"""Input forms through the class API: dtypes, memory orders, views, DataFrames -> the same graph as the float32 C array; what the reference rejects
(and HOW: its constructor reads data.shape before check_array, pynndescent_.py:1009-1036) is rejected the same way.  usage: input_forms.py"""
import os, sys, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, pandas as pd
from pynndescent_amd import NNDescent
rs = np.random.RandomState(0)
base = rs.standard_normal((3000, 20)).astype(np.float32)
ref = NNDescent(base, "euclidean", n_neighbors=10, random_state=1).neighbor_graph
forms = {
    "float64": base.astype(np.float64),
    "fortran": np.asfortranarray(base),
    "strided view": np.repeat(base, 2, axis=0)[::2],
    "column-sliced view": np.hstack([base, base])[:, :20],
    "read-only": (lambda a: (a.setflags(write=False), a)[1])(base.copy()),
    "DataFrame": pd.DataFrame(base),
    "int32 (cast)": (base * 100).astype(np.int32),
}
bad = 0
for name, x in forms.items():
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx, dist = NNDescent(x, "euclidean", n_neighbors=10, random_state=1).neighbor_graph
        if name.startswith("int32"):
            ok = idx.shape == (3000, 10)
        else:
            ok = np.array_equal(idx, ref[0]) and np.allclose(dist, ref[1], rtol=1e-6)
        print("%-20s %s" % (name, "ok" if ok else "DIFFERS"))
        bad += 0 if ok else 1
    except Exception as e:
        print("%-20s EXC %s: %s" % (name, type(e).__name__, str(e)[:100])); bad += 1
for name, x, exc in [("NaN", np.where(rs.rand(3000, 20) < 0.001, np.nan, base).astype(np.float32), ValueError), ("inf", np.where(rs.rand(3000, 20) < 0.001, np.inf, base).astype(np.float32), ValueError),
                     # the reference reads data.shape[0] / data.shape[1] before check_array (P_:1009-1036): these are ITS exceptions
                     ("list of lists", base.tolist(), AttributeError), ("1-d", base[:, 0], IndexError), ("empty", base[:0], OverflowError),
                     ("3-d", base.reshape(3000, 4, 5), ValueError)]:
    try:
        NNDescent(x, "euclidean", n_neighbors=10, random_state=1)
        print("%-20s accepted (expected %s)" % (name, exc.__name__)); bad += 1
    except exc as e:
        print("%-20s raises %s: %s" % (name, type(e).__name__, str(e)[:80]))
    except Exception as e:
        print("%-20s EXC %s: %s" % (name, type(e).__name__, str(e)[:100])); bad += 1
print("bad:", bad)
sys.exit(1 if bad else 0)

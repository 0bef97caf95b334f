import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from pynndescent_amd import sharded, NNDescent
from tests.util_data import clustered
x = clustered(60_000, 48, 10, 120, seed=41)
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
print("first", flush=True)
sharded.build_multi(x, 2, devices=[0, 0], n_neighbors=15, n_trees=8, seed=3)
if mode == "b":
    print("single", flush=True)
    NNDescent(x, "euclidean", n_neighbors=15, n_trees=8, random_state=3)
print("second", flush=True)
sharded.build_multi(x, 2, devices=[0, 0], n_neighbors=15, n_trees=8, seed=3)
print("third (class)", flush=True)
NNDescent(x, "euclidean", n_neighbors=15, n_trees=8, random_state=3, n_devices=2, devices=[0, 0])
print("ok", flush=True)

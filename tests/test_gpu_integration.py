"""The drop-in boundary as a reference maintainer would use it: INTEGRATION.md section 2's ctypes stub, executed
verbatim (host buffers through ``nnd_build``), and bench.py's multi-process launch path."""
import json
import os
import re
import subprocess
import time
import sys

import numpy as np
import pytest

from oracle import oracle as O
from tests.util_data import clustered

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _integration_stub():
    """The python code block of INTEGRATION.md section 2, with the library name resolved to the in-tree build."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    code = re.search(r"```python\n(.*?)```", sec, re.DOTALL).group(1)
    lib = os.path.join(ROOT, "pynndescent_amd", "libpynnd_amd.so")
    assert 'C.CDLL("libpynnd_amd.so")' in code
    code = code.replace('C.CDLL("libpynnd_amd.so")', "C.CDLL(%r)" % lib)
    ns = {}
    exec(compile(code, "INTEGRATION.md#2", "exec"), ns)
    return ns["_gpu_nn_descent"]


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
def test_integration_md_stub_verbatim(metric):
    """nnd_build -- the ONE function the stub binds -- with numpy buffers in and out, against the CPU oracle."""
    gpu_nn_descent = _integration_stub()
    x = clustered(20000, 48, 10, 64, seed=3)
    k, n_trees, seed = 15, 8, 11
    rng_state, _, tree_states = O.draw_rng_states(seed, n_trees)
    idx, dist = gpu_nn_descent(x, k, rng_state, tree_states, min(60, k), metric, O.default_n_iters(x.shape[0]), 0.001,
                               n_trees, O.default_leaf_size(k), 200)
    assert idx.shape == dist.shape == (20000, k) and idx.dtype == np.int32 and dist.dtype == np.float32
    assert (idx >= 0).all() and np.all(np.diff(dist, axis=1) >= 0)
    assert np.mean(idx[:, 0] == np.arange(20000)) > 0.999
    oidx, _ = O.build_index(x, metric, n_neighbors=k, n_trees=n_trees, random_state=seed, n_threads=16, kind="fast")
    ti, _ = O.brute_force_knn(x, 10, metric)
    r_gpu, r_cpu = O.recall(ti, idx), O.recall(ti, oidx)
    print("INTEGRATION.md stub (%s): recall@10 gpu %.4f oracle %.4f" % (metric, r_gpu, r_cpu))
    assert abs(r_gpu - r_cpu) <= 0.005
    # alt-space distances, exact for the returned ids
    xi = x.astype(np.float64)
    if metric == "euclidean":
        np.testing.assert_allclose(dist, ((xi[:, None, :] - xi[idx]) ** 2).sum(-1), rtol=1e-5, atol=1e-6)


def test_integration_md_stub_two_devices():
    """The stub's n_devices argument (the GPU analogue of n_jobs, pynndescent_.py:1141-1143): nnd_build_multi, two ranks --
    both on this box's one GPU (devices=[0, 0]: the library's LOCAL transport; distinct ordinals would talk over RCCL)."""
    gpu_nn_descent = _integration_stub()
    x = clustered(40000, 48, 10, 64, seed=3)
    k, n_trees, seed = 15, 8, 11
    rng_state, _, tree_states = O.draw_rng_states(seed, n_trees)
    args = (x, k, rng_state, tree_states, min(60, k), "euclidean", O.default_n_iters(x.shape[0]), 0.001, n_trees, O.default_leaf_size(k), 200)
    idx2, dist2 = gpu_nn_descent(*args, n_devices=2, devices=[0, 0])
    idx1, _ = gpu_nn_descent(*args)
    assert idx2.shape == (40000, k) and (idx2 >= 0).all() and np.all(np.diff(dist2, axis=1) >= 0)
    rows = np.arange(0, 40000, 8)
    ti, _ = O.brute_force_knn(x, 10, "euclidean", rows=rows, kind="fast")
    r2, r1 = O.recall(ti, idx2[rows]), O.recall(ti, idx1[rows])
    print("INTEGRATION.md stub: recall@10 with n_devices=2 %.4f, one device %.4f" % (r2, r1))
    assert abs(r2 - r1) <= 0.005
    xi = x[rows].astype(np.float64)
    np.testing.assert_allclose(dist2[rows], ((xi[:, None, :] - x[idx2[rows]].astype(np.float64)) ** 2).sum(-1), rtol=1e-5, atol=1e-6)


def test_integration_md_stub_init_graph_and_error():
    gpu_nn_descent = _integration_stub()
    x = clustered(3000, 16, 5, 20, seed=5)
    ti, _ = O.brute_force_knn(x, 10, "euclidean")
    rs = np.random.RandomState(0)
    noisy = np.where(rs.uniform(size=ti.shape) < 0.5, rs.randint(0, 3000, ti.shape), ti).astype(np.int32)
    rng_state, _, tree_states = O.draw_rng_states(1, 1)
    idx, _ = gpu_nn_descent(x, 10, rng_state, tree_states, 10, "euclidean", 12, 0.001, 4, 60, 200, init_graph=noisy)
    assert O.recall(ti, idx) > 0.95
    with pytest.raises(RuntimeError, match="n_neighbors"):  # no silent fallback: the library's message surfaces
        gpu_nn_descent(x, 300, rng_state, tree_states, 60, "euclidean", 12, 0.001, 4, 60, 200)


def test_bench_two_processes_sharing_one_gpu():
    """`python bench.py --gpus 2` with no launcher: the script must re-execute itself as 2 ranks (here both on GPU 0
    over gloo -- PYNND_BENCH_SHARE_GPU=1 -- since the test box has one GPU) and report n_gpus = 2."""
    env = dict(os.environ, PYNND_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--points-per-gpu", "60000",
                          "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["recall_at_10"] >= 0.95
    assert sum(rec["exchanged_records_rank0"]) > 0


def test_bench_eight_processes_rehearsal_of_the_drivers_scaling_run():
    """What the driver runs for SCALE at N = 8 -- `bench.py --gpus 8`, eight processes, torch.distributed rendezvous, the
    by-cell forest, every exchange of the sharded build, the watchdog armed -- rehearsed on this one-GPU box: all ranks on
    GPU 0, the HOST transport over gloo (two RCCL ranks cannot share a GPU).  300 k points per rank (the configs[3] run is
    1.25 M per rank; the eight thread-ranks of tests/test_gpu_sharded.py cover that size in one process)."""
    env = dict(os.environ, PYNND_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--points-per-gpu", "300000", "--n-trees", "8",
                          "--steps", "1", "--warmup", "1", "--no-one-gpu"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert "error" not in rec, rec
    assert rec["n_gpus"] == 8 and rec["value"] > 0 and rec["recall_at_10"] >= 0.95
    assert rec["config"]["comm"]["transport"] == "host" and rec["shard_rank0"]["forest_by_cell"]
    print("bench.py --gpus 8 (8 processes, one GPU, HOST transport): %.1f ms per build of %d points, recall@10 %.4f"
          % (rec["ms_per_step"], 8 * 300000, rec["recall_at_10"]))


def test_bench_watchdog_prints_a_line_when_a_rank_never_arrives():
    """A rank that never enters its build (hook: it sleeps) must not leave the run hanging without a line: rank 0's
    watchdog prints the contract's fields with `value` 0 and an `error` naming the phase, and the run ends."""
    env = dict(os.environ, PYNND_BENCH_SHARE_GPU="1", PYNND_BENCH_TEST_STALL_RANK="1", PYNND_BENCH_WATCHDOG_SCALE="0.03")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--points-per-gpu", "60000", "--steps", "1", "--warmup", "1",
                          "--no-one-gpu"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert lines, (out.stdout[-1000:], out.stderr[-2000:])
    rec = json.loads(lines[-1])
    assert rec["value"] == 0.0 and rec["n_gpus"] == 2 and "warm-up build" in rec["error"], rec
    assert time.time() - t0 < 240


def test_bench_refuses_more_gpus_than_visible():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PYNND_BENCH_SHARE_GPU"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode != 0 and "--gpus 64 requested" in (out.stderr + out.stdout)

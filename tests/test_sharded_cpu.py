"""Host side of the row-sharded build, on CPU: the partition helper, the byte-level exchange callback the library's HOST
transport calls (csrc/comm.hip) over gloo with two and three processes, the parameter draw order, bench.py's loaders."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from pynndescent_amd import sharded


def test_partition_helpers():
    r = sharded.shard_ranges(10, 3)
    assert r == [(0, 3), (3, 6), (6, 10)]
    assert sharded.shard_ranges(8, 8) == [(i, i + 1) for i in range(8)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import ctypes as C

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the HOST transport's callback (nnd_host_exchange_fn): all-to-all-v of byte segments of host buffers, the way
        # csrc/comm.hip calls it -- segments of different lengths and empty ones
        cb = sharded._host_callback(dist)
        lens = [[(3 * s + 5 * d) % 7 for d in range(world)] for s in range(world)]  # lens[src][dst] bytes
        lens[0][1] = 0
        sb = np.concatenate([np.full(lens[rank][dst], 10 * rank + dst, np.uint8) for dst in range(world)] + [np.zeros(0, np.uint8)])
        rb = np.zeros(sum(lens[src][rank] for src in range(world)), np.uint8)
        i64 = lambda v: (C.c_int64 * world)(*[int(q) for q in v])  # noqa: E731
        so = np.concatenate([[0], np.cumsum(lens[rank])[:-1]])
        ro = np.concatenate([[0], np.cumsum([lens[src][rank] for src in range(world)])[:-1]])
        rc_ = cb(None, sb.ctypes.data if sb.size else 1, i64(so), i64(lens[rank]), rb.ctypes.data if rb.size else 1, i64(ro),
                 i64([lens[src][rank] for src in range(world)]))
        want = np.concatenate([np.full(lens[src][rank], 10 * src + rank, np.uint8) for src in range(world)] + [np.zeros(0, np.uint8)])
        ok = rc_ == 0 and np.array_equal(rb, want)
        # a data exchange in which ONE rank moves no byte must not turn into a barrier on that rank (with three ranks the
        # other two are in point-to-point calls): rank 2 sends and receives nothing, ranks 0 and 1 swap a byte
        z = [0] * world
        if rank == 2:
            ok = ok and cb(None, sb.ctypes.data if sb.size else 1, i64(z), i64(z), rb.ctypes.data if rb.size else 1, i64(z), i64(z)) == 0
        else:
            peer = 1 - rank
            one = np.array([40 + rank], np.uint8)
            got = np.zeros(1, np.uint8)
            cnt = [1 if q == peer else 0 for q in range(world)]
            ok = ok and cb(None, one.ctypes.data, i64(z), i64(cnt), got.ctypes.data, i64(z), i64(cnt)) == 0 and int(got[0]) == 40 + peer
        # the transport's barrier: send == recv == NULL
        ok = ok and cb(None, None, i64(z), i64(z), None, i64(z), i64(z)) == 0
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_host_exchange_callback_gloo(world):
    if world == 3 and (os.cpu_count() or 1) < 3:
        pytest.skip("needs three processes")
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_global_params_follow_the_reference_draw_order():
    """The sharded build's parameters (identical on every rank) use the RandomState draw order of the class
    (pynndescent_.py:1105-1113: rng_state, search_rng_state, then make_forest's per-tree states, rp_trees.py:2850) and the
    reference's derived defaults as the oracle restates them (leaf size, max_candidates, n_iters)."""
    from oracle import oracle as O

    p = sharded._global_params(123456, 64, "cosine", 20, 6, None, None, None, 0.001, 77, 200, 0)
    rng_state, _, tree_states = O.draw_rng_states(77, 6)
    assert [p.rng_state[i] for i in range(3)] == [int(v) for v in rng_state]
    assert [p.tree_rng[i] for i in range(3)] == [int(v) for v in tree_states[0]]
    assert p.n == 123456 and p.dim == 64 and p.n_neighbors == 20 and p.n_trees == 6
    assert p.leaf_size == O.default_leaf_size(20) and p.max_candidates == 20
    assert p.n_iters == O.default_n_iters(123456) and abs(p.delta - 0.001) < 1e-9


def test_bench_load_points_formats(tmp_path):
    """bench.py --data: .npy and TEXMEX .fvecs (int32 d, then d floats, per row) give the same float32 (n, d) array."""
    import bench

    x = np.random.RandomState(0).standard_normal((37, 12)).astype(np.float32)
    np.save(tmp_path / "pts.npy", x.astype(np.float64))  # any dtype on disk: converted
    rows = np.empty((37, 13), np.int32)
    rows[:, 0] = 12
    rows[:, 1:] = x.view(np.int32)
    rows.tofile(tmp_path / "pts.fvecs")
    a = bench.load_points(str(tmp_path / "pts.npy"))
    b = bench.load_points(str(tmp_path / "pts.fvecs"))
    assert a.dtype == b.dtype == np.float32 and a.flags.c_contiguous and b.flags.c_contiguous
    np.testing.assert_array_equal(a, x)
    np.testing.assert_array_equal(b, x)

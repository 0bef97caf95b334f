"""Host side of the row-sharded build, on CPU: partition helpers, the torch.distributed tensor transport (gloo,
world_size 2) and the byte-level exchange callback the library's HOST transport calls (csrc/comm.hip)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from pynndescent_amd import sharded


def test_partition_helpers():
    r = sharded.shard_ranges(10, 3)
    assert r == [(0, 3), (3, 6), (6, 10)]
    assert sharded.shard_ranges(8, 8) == [(i, i + 1) for i in range(8)]
    assert sharded.tree_ranges(8, 2) == [(0, 4), (4, 8)]
    tr = sharded.tree_ranges(3, 8)  # fewer trees than ranks: some ranks get none, every tree is built once
    assert sum(b - a for a, b in tr) == 3 and all(b - a in (0, 1) for a, b in tr)
    # records are ordered by target vertex: the exclusive scan at the range edges gives the per-rank segments
    cnt = np.array([0, 2, 0, 1, 0, 0, 3, 1, 0, 0])
    off = np.concatenate([[0], np.cumsum(cnt)])
    seg = sharded.segment_bounds(off, r)
    assert seg == [(0, 2), (2, 3), (3, 7)]
    assert sum(b - a for a, b in seg) == cnt.sum()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = sharded.TorchDistComm()
        # all-gather of shards with different row counts (k-list rows / point-set shards)
        mine = torch.full((3 + rank, 4), float(rank))
        got = comm.all_gather_v(mine)
        ok = len(got) == world and all(g.shape == (3 + r, 4) and bool((g == r).all()) for r, g in enumerate(got))
        # all-to-all-v with ragged (and empty) segments: the proposal records
        send = [torch.arange(rank * 100 + dst * 10, rank * 100 + dst * 10 + (dst + rank) % 3, dtype=torch.int64)
                for dst in range(world)]
        recv, rc = comm.all_to_all_v(send, return_counts=True)
        ok = ok and rc == [(rank + src) % 3 for src in range(world)]
        # a second array with the same segmentation reuses the counts (no second count exchange)
        recv2 = comm.all_to_all_v([t.to(torch.int32) * 2 for t in send], rcounts=rc)
        ok = ok and all(torch.equal(a.to(torch.int32) * 2, b) for a, b in zip(recv, recv2))
        for src in range(world):
            n = (rank + src) % 3
            want = torch.arange(src * 100 + rank * 10, src * 100 + rank * 10 + n, dtype=torch.int64)
            ok = ok and torch.equal(recv[src], want)
        # the HOST transport's callback (nnd_host_exchange_fn): all-to-all-v of byte segments of host buffers, the way
        # csrc/comm.hip calls it -- segments of different lengths, an empty one, and the zero-byte call (barrier)
        import ctypes as C

        cb = sharded._host_callback(comm)
        lens = [[3, 5], [0, 7]]  # lens[src][dst] bytes
        sb = np.concatenate([np.full(lens[rank][dst], 10 * rank + dst, np.uint8) for dst in range(world)])
        rb = np.zeros(sum(lens[src][rank] for src in range(world)), np.uint8)
        i64 = lambda v: (C.c_int64 * world)(*v)  # noqa: E731
        so = np.concatenate([[0], np.cumsum(lens[rank])[:-1]])
        ro = np.concatenate([[0], np.cumsum([lens[src][rank] for src in range(world)])[:-1]])
        rc_ = cb(None, sb.ctypes.data, i64(so), i64(lens[rank]), rb.ctypes.data, i64(ro), i64([lens[src][rank] for src in range(world)]))
        want = np.concatenate([np.full(lens[src][rank], 10 * src + rank, np.uint8) for src in range(world)])
        ok = ok and rc_ == 0 and np.array_equal(rb, want)
        ok = ok and cb(None, sb.ctypes.data, i64([0] * world), i64([0] * world), rb.ctypes.data, i64([0] * world), i64([0] * world)) == 0
        # update-count all-reduce (stop rule, pynndescent_.py:317)
        ok = ok and comm.all_reduce_sum(5 + rank) == sum(5 + r for r in range(world))
        comm.barrier()
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_torch_dist_transport_gloo(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_thread_comm_matches_the_contract():
    import threading

    comms = sharded.ThreadComm.make(3)
    res = [None] * 3

    def run(r):
        c = comms[r]
        g = c.all_gather_v(torch.full((r + 1,), r))
        a = c.all_to_all_v([torch.tensor([r * 10 + d] * ((r + d) % 2)) for d in range(3)])
        s = c.all_reduce_sum(r + 1)
        res[r] = (g, a, s)

    ts = [threading.Thread(target=run, args=(r,)) for r in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for r in range(3):
        g, a, s = res[r]
        assert [t.tolist() for t in g] == [[0], [1, 1], [2, 2, 2]]
        assert [t.tolist() for t in a] == [[src * 10 + r] * ((src + r) % 2) for src in range(3)]
        assert s == 6


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

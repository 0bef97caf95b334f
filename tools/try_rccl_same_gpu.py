"""Can two RCCL ranks live on ONE GPU (would let the one-GPU test box exercise the RCCL transport)?  Prints what happens."""
import ctypes as C, os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp


def worker(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pynndescent_amd import _capi
    lib = _capi.load_library()
    ident = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_uint8 * 128)()
        print("unique id rc", lib.nnd_comm_unique_id(buf), flush=True)
        ident = torch.tensor(list(buf), dtype=torch.uint8)
    dist.broadcast(ident, 0)
    h = _capi._H()
    rc = lib.nnd_comm_create_rccl(C.byref(h), bytes(ident.tolist()), world, rank, 0)
    print("rank", rank, "create_rccl rc", rc, lib.nnd_comm_last_error(None).decode() if rc else "ok", flush=True)
    dist.barrier()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(world, port), nprocs=world, join=True)

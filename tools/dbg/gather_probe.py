"""Probe: all_gather_into_tensor / all_to_all_single on device tensors under an external stream, gloo with two ranks on
one GPU (rehearsal backend) and RCCL with one rank. Usage: python tools/dbg/gather_probe.py gloo|nccl"""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, backend, port):
    faulthandler.dump_traceback_later(40, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    st = torch.cuda.Stream(device=dev)
    ext = torch.cuda.ExternalStream(st.cuda_stream, device=dev)
    for nq in (1000, 23531, 94000):
        for mode in ("default", "external"):
            local = torch.full((nq, 4), rank, dtype=torch.int32, device=dev)
            lists = torch.zeros((world * nq, 4), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            t0 = time.time()
            if mode == "external":
                with torch.cuda.stream(ext):
                    dist.all_gather_into_tensor(lists, local)
            else:
                dist.all_gather_into_tensor(lists, local)
            torch.cuda.synchronize()
            ok = all(int(lists[r * nq, 0]) == r for r in range(world))
            print("[%d] %s all_gather nq=%d %s ok=%s %.3fs" % (rank, backend, nq, mode, ok, time.time() - t0), flush=True)
    # in-place gather (input = the rank's slot of the output) and all_to_all
    cap = 4096
    qall = torch.zeros((world, cap, 128), dtype=torch.uint8, device=dev)
    qall[rank] = rank + 1
    with torch.cuda.stream(ext):
        dist.all_gather_into_tensor(qall.view(world * cap, 128), qall[rank] if backend == "nccl" else qall[rank].clone())
        a = torch.full((world * cap, 4), rank, dtype=torch.int32, device=dev)
        b = torch.zeros((world * cap, 4), dtype=torch.int32, device=dev)
        dist.all_to_all_single(b, a)
    torch.cuda.synchronize()
    print("[%d] in-place gather ok=%s all_to_all ok=%s" % (rank, all(int(qall[r, 0, 0]) == r + 1 for r in range(world)),
                                                         all(int(b[r * cap, 0]) == r for r in range(world))), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    backend = sys.argv[1]
    world = 2 if backend == "gloo" else 1
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(world, backend, port), nprocs=world, join=True)

#!/usr/bin/env python3
"""Does RCCL accept two ranks on ONE GPU?  (It would let the pipelined `rccl` exchange be
rehearsed over the real backend on the one-GPU boxes this project is built on.)

    python tools/rccl_same_gpu_probe.py"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    try:
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
        x = torch.full((4,), float(rank), device='cuda')
        out = torch.empty(4 * world, device='cuda')
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        print(rank, 'all_gather ok', out.tolist(), flush=True)
        buf = torch.zeros(8, device='cuda')
        ops = [dist.P2POp(dist.isend, x.repeat(2), 0)] if rank else [dist.P2POp(dist.irecv, buf, 1)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        print(rank, 'p2p ok', buf.tolist(), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print(rank, 'FAILED', repr(e)[:600], flush=True)
        os._exit(0)


if __name__ == '__main__':
    mp.spawn(worker, args=(2, 29611), nprocs=2, join=True)

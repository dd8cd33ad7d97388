#!/usr/bin/env python3
"""The torch.distributed calls of the pipelined `rccl` exchange (dist._trace_spot_pipelined) over
the REAL backend with the one rank a one-GPU box allows: the per-stage count all-gather of a
slice on a side stream, its copy to pinned memory behind it, grouped isend / irecv of row slices
of 2-D tensors (to itself: RCCL runs a send and a receive to the same rank inside one group),
an event for the copy stream.  RCCL refuses two ranks on one GPU ("Duplicate GPU detected"),
so this is as much of the NCCL side as can run here; the multi-rank logic runs over gloo.

    python tools/rccl_self_p2p_probe.py"""
import json
import os

import torch
import torch.distributed as dist


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29612')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    side, copy = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    S, n = 4, 1 << 20
    xy = torch.randn((S * n, 2), dtype=torch.float64, device=dev)
    stage = torch.zeros_like(xy)
    cnt = torch.arange(1, S + 1, dtype=torch.int64, device=dev) * 1000
    allc_dev = torch.zeros((S, 1), dtype=torch.int64, device=dev)
    allc = torch.zeros((S, 1), dtype=torch.int64).pin_memory()
    ok = True
    for s in range(S):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        side.wait_event(ev)
        with torch.cuda.stream(side):
            dist.all_gather_into_tensor(allc_dev[s], cnt[s:s + 1])
            allc[s].copy_(allc_dev[s], non_blocking=True)
            ev_c = torch.cuda.Event()
            ev_c.record(side)
        ev_c.synchronize()
        k = int(allc[s, 0])
        ok &= k == (s + 1) * 1000
        with torch.cuda.stream(side):
            ops = [dist.P2POp(dist.irecv, stage[s * n:s * n + k], 0),
                   dist.P2POp(dist.isend, xy[s * n:s * n + k], 0)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            ev_x = torch.cuda.Event()
            ev_x.record(side)
        copy.wait_event(ev_x)
    copy.synchronize()
    side.synchronize()
    for s in range(S):
        k = (s + 1) * 1000
        ok &= bool(torch.equal(stage[s * n:s * n + k], xy[s * n:s * n + k]))
        ok &= bool((stage[s * n + k:(s + 1) * n] == 0).all())
    print(json.dumps({'backend': dist.get_backend(), 'stages': S, 'self_p2p_and_count_gather_ok': ok}))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

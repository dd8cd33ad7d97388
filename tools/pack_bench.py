#!/usr/bin/env python3
"""Packed-hits kernels of a whole configuration vs its plain HITS kernels.

    [ROX_LIB=variant.so] python tools/pack_bench.py [--workload litho_c5] [--num 1024]

One pass = every (field, wavelength) grid: (a) HITS launches into one reused buffer,
(b) dist.trace_blocks -- HITS_COMPACT | HITS_APPEND launches packing into one HBM buffer.
Steady-state ms per pass from events on the launch stream."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='litho_c5')
    ap.add_argument('--num', type=int, default=1024)
    ap.add_argument('--passes', type=int, default=5)
    args = ap.parse_args()
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads, engine, dist as rdist
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    wl = workloads.load(args.workload)
    N, num = wl.n_ifcs, args.num
    eng = TraceEngine(wl.table)
    nf, nw = len(wl.fields), len(wl.table.wvls)
    plan = rdist.partition(nf, nw, num, 1)
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    hits = DeviceResult(torch, eng.device, 0, num * num, abi.OUT_HITS, want_pupil=False, nan_fill=False)
    grid = make_grid((-1., -1.), (1., 1.), num)

    def pass_hits():
        for b in plan[0]:
            o = make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2, foc=wl.foc,
                          image_pt=wl.image_pts[b.fi])
            eng.trace_pupil_grid(wl.fields[b.fi], grid, b.wi, o, want_pupil=False, out=hits)

    def pass_pack():
        return rdist.trace_blocks(eng, plan[0], num, wl.fields, wl.image_pts, wl.foc)

    def timed(fn):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            fn()
            torch.cuda.synchronize()
        ts = []
        for _ in range(args.passes):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
            del r
        return sorted(ts)[len(ts) // 2]
    res = {'lib': os.path.basename(engine.LIB_PATH), 'workload': args.workload, 'num': num,
           'grids': len(plan[0]), 'rays': rdist.rays_of(plan[0], num)}
    res['hits_ms'] = timed(pass_hits)
    res['pack_ms'] = timed(pass_pack)
    res['pack_over_hits'] = res['pack_ms'] / res['hits_ms']
    print(json.dumps(res))


if __name__ == '__main__':
    main()

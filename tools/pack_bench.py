#!/usr/bin/env python3
"""Packed-hits kernels of a whole configuration vs its plain HITS kernels.

    [ROX_LIB=variant.so] python tools/pack_bench.py [--workload litho_c5] [--num 1024]

One pass = every (field, wavelength) grid: (a) HITS launches into one reused buffer,
(b) dist.trace_blocks -- HITS_COMPACT | HITS_APPEND launches packing into one HBM buffer --
in both forms of the library: fused (compaction inside the trace kernel, ROX_PACK_TWO_PASS=0)
and two-pass (plain HITS launch + csrc/pack.hip, ROX_PACK_TWO_PASS=1), and what the library's
own rule picks.  Steady-state ms per pass from events on the launch stream.
--all runs the crossover table (profiles/r04_pack_crossover.jsonl)."""
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
    ap.add_argument('--all', action='store_true')
    args = ap.parse_args()
    if args.all:
        for name, num in (('dblgauss_c2', 64), ('dblgauss_c2', 128), ('dblgauss_c2', 256), ('dblgauss_c2', 512),
                          ('litho_c5', 128), ('litho_c5', 256), ('singlet_c1', 1024), ('rc_telescope_c4', 1024), ('tt_triplet', 1024),
                          ('dblgauss_c2', 1024), ('zmx_evenasph_c3', 1024), ('cell_phone', 1024),
                          ('nikkor_c3', 1024), ('litho_c5', 1024), ('litho_c5', 2048)):
            args.workload, args.num = name, num
            one(args)
        return
    one(args)


def one(args):
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
    res['interfaces'] = N
    res['hits_ms'] = timed(pass_hits)
    os.environ['ROX_PACK_TWO_PASS'] = '0'
    res['fused_ms'] = timed(pass_pack)
    os.environ['ROX_PACK_TWO_PASS'] = '1'
    res['two_pass_ms'] = timed(pass_pack)
    os.environ['ROX_PACK_TWO_PASS'] = ''
    import ctypes as C
    c0 = (C.c_uint64 * 2)()
    c1 = (C.c_uint64 * 2)()
    eng.lib.rox_diag_pack_launches(c0)
    res['auto_ms'] = timed(pass_pack)
    eng.lib.rox_diag_pack_launches(c1)
    res['auto_picks'] = 'two_pass' if c1[1] > c0[1] else 'fused'
    res['two_pass_over_fused'] = res['two_pass_ms'] / res['fused_ms']
    res['fused_over_hits'] = res['fused_ms'] / res['hits_ms']
    res['two_pass_over_hits'] = res['two_pass_ms'] / res['hits_ms']
    print(json.dumps(res), flush=True)
    eng.close()
    del hits
    torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

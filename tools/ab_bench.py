#!/usr/bin/env python3
"""A/B harness for kernel experiments on the GPU box.

    ROX_LIB=/path/to/variant.so python tools/ab_bench.py [--num 1024] [--check]

Times the FULL and HITS pupil-grid kernels of BASELINE.json configs[1] with HIP
events (rox_time_pupil_grid), interleaved repeats, and optionally checks a
128x128 grid bit-for-bit against the oracle."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--num', type=int, default=1024)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--launches', type=int, default=20)
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--workload', default='dblgauss_c2')
    ap.add_argument('--field', type=int, default=0)
    ap.add_argument('--ld-pad', type=int, default=-1, help='row pitch = R + pad doubles (-1: engine default)')
    args = ap.parse_args()
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads, engine
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    wl = workloads.load(args.workload)
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    fld = wl.fields[args.field]
    wi = wl.ref_wvl_idx
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    if fld.kind == abi.FLD_EPD_WIDE or fld.z_dir0 == 0.0:
        flags &= ~abi.INTERSECT_OBJ             # wide-angle fields (trace.py:302-303)
    grid = make_grid((-1., -1.), (1., 1.), args.num)
    R = args.num ** 2
    res = {'lib': os.path.basename(engine.LIB_PATH)}
    if args.check:
        from oracle import oracle
        g = make_grid((-1., -1.), (1., 1.), 128)
        for mode in (abi.OUT_FULL, abi.OUT_HITS):
            o = make_opts(flags=flags, out_mode=mode, first_surf=1, last_surf=N - 2,
                          foc=0.01, image_pt=wl.image_pts[args.field])
            dev = eng.trace_pupil_grid(fld, g, wi, o, nan_fill=True).to_host()
            orc = oracle.trace_pupil_grid(wl.table, fld, g, wi, o)
            same = (np.array_equal(dev.seg, orc.seg, equal_nan=True)
                    and np.array_equal(dev.op, orc.op, equal_nan=True)
                    and np.array_equal(dev.status, orc.status)
                    and np.array_equal(dev.fail_surf, orc.fail_surf))
            res[f'bit_exact_mode{mode}'] = bool(same)
            if not same:
                m = ~(np.isnan(dev.seg) & np.isnan(orc.seg))
                res[f'maxdiff_mode{mode}'] = float(np.nanmax(np.abs(dev.seg[m] - orc.seg[m])))
    outs = {}
    for name, mode in (('full', abi.OUT_FULL), ('hits', abi.OUT_HITS), ('hits_fast', abi.OUT_HITS)):
        o = make_opts(flags=flags | (abi.FAST_FP64 if name == 'hits_fast' else 0), out_mode=mode,
                      first_surf=1, last_surf=N - 2,
                      foc=wl.foc, image_pt=wl.image_pts[args.field])
        out = DeviceResult(torch, eng.device, eng.num_segments(flags), R, mode,
                           want_pupil=(mode == abi.OUT_FULL), nan_fill=False,
                           ld=None if args.ld_pad < 0 else R + args.ld_pad)
        outs[name] = (o, out)
    # each mode measured at steady-state clocks (120 ms of launches first: a cold
    # batch reads 10-30 % slow and A/B differences drown in the clock ramp)
    times = {k: [] for k in outs}
    for name, (o, out) in outs.items():
        eng.time_pupil_grid_sustained(fld, grid, wi, o, out, args.launches, 1)
        for _ in range(args.reps):
            times[name].append(eng.time_pupil_grid(fld, grid, wi, o, out, args.launches))
    for k, v in times.items():
        res[k + '_us'] = round(float(np.median(v)) * 1e3, 2)
        res[k + '_min_us'] = round(float(np.min(v)) * 1e3, 2)
    # work of one launch (for per-intersection counter figures, tools/make_valu.py)
    st = outs['hits'][1].status
    fs = outs['hits'][1].fail_surf.to(torch.int64)
    ok = st == abi.OK
    res.update(workload=args.workload, field=args.field, num=args.num, rays=R,
               rays_through=int(ok.sum().item()),
               intersections=int(ok.sum().item()) * (N - 1) + int(fs[~ok].sum().item()))
    print(json.dumps(res))


if __name__ == '__main__':
    main()

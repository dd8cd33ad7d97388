#!/usr/bin/env python3
"""Spot-diagram wall-clock at the product boundary, and what it is made of.

    python tools/spot_wallclock.py [--num 1024] [--reps 30] [--workload dblgauss_c2] [--field 0]

A  trace.trace_grid_spot on a table-backed model: Python call -> host (R_ok, 2)
   array; the kernel packs the survivors and writes them into pinned host memory
B  same kernel into an HBM buffer, then count -> D2H of n*16 bytes into pinned
C  round 1's shape: HITS kernel + torch nonzero / index_select / stack + D2H
plus kernel-only durations (HIP events) of HITS, HITS_COMPACT->HBM, HITS_COMPACT->pinned."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--num', type=int, default=1024)
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--workload', default='dblgauss_c2')
    ap.add_argument('--field', type=int, default=0)
    args = ap.parse_args()
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads, trace, session
    from rayoptics_amd.engine import make_opts, make_grid, DeviceResult, load_library

    model = workloads.TableModel(args.workload)
    wl = model.workload
    fld = model.fields[args.field]
    wvl = wl.table.wvls[wl.ref_wvl_idx]
    num = args.num
    R = num * num
    grid_rng = [np.array([-1., -1.]), np.array([1., 1.]), num]
    image_pt = wl.image_pts[args.field]
    res = {'workload': args.workload, 'num': num, 'rays': R}

    def timed(fn, reps):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return out, float(np.median(ts)), float(np.min(ts))

    # A: the product function
    xy = trace.trace_grid_spot(model, grid_rng, fld, wvl, wl.foc, image_pt)     # warm
    xy, med, mn = timed(lambda: trace.trace_grid_spot(model, grid_rng, fld, wvl, wl.foc, image_pt),
                        args.reps)
    res['A_product_pinned_ms'] = {'median': med, 'min': mn, 'rays_through': int(xy.shape[0])}
    ref_xy = xy.copy()
    # A': every wavelength of the field -- one call per wavelength (SequentialModel.trace_grid's
    # loop) against trace_grid_spots (one batched launch, one synchronise)
    wvls = list(wl.table.wvls)
    if len(wvls) > 1:
        def looped():
            return [trace.trace_grid_spot(model, grid_rng, fld, w, wl.foc, image_pt) for w in wvls]
        trace.trace_grid_spots(model, grid_rng, fld, wvls, wl.foc, image_pt)            # warm
        one, med1, mn1 = timed(looped, args.reps)
        allw, med2, mn2 = timed(lambda: trace.trace_grid_spots(model, grid_rng, fld, wvls, wl.foc, image_pt),
                                args.reps)
        assert all(np.array_equal(a, b) for a, b in zip(one, allw))
        res['A_all_wavelengths_ms'] = {'wavelengths': len(wvls), 'one_call_per_wavelength': {'median': med1, 'min': mn1},
                                       'one_batched_launch': {'median': med2, 'min': mn2}}
        del one, allw

    eng = session.engine_for(model)
    lib = load_library()
    N = wl.n_ifcs
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    f = fld.rox_field
    wi = wl.ref_wvl_idx
    grid = make_grid((-1., -1.), (1., 1.), num)
    o_cmp = make_opts(flags=flags, out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2,
                      foc=wl.foc, image_pt=image_pt)
    o_hits = make_opts(flags=flags, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                       foc=wl.foc, image_pt=image_pt)

    # B: compaction into HBM + exact-size D2H
    xy_d = torch.empty((R, 2), dtype=torch.float64, device=eng.device)
    n_pin = torch.zeros(1, dtype=torch.int64).pin_memory()
    xy_pin = torch.empty((R, 2), dtype=torch.float64).pin_memory()
    out_b = abi.Out()
    out_b.seg, out_b.n_hits, out_b.ld = xy_d.data_ptr(), n_pin.data_ptr(), R

    def run_b():
        rc = lib.rox_trace_pupil_grid(eng._handle, C.byref(f), C.byref(grid), wi, C.byref(o_cmp),
                                      C.byref(out_b), eng._stream())
        assert rc == 0
        torch.cuda.current_stream().synchronize()
        n = int(n_pin[0])
        xy_pin[:n].copy_(xy_d[:n], non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return xy_pin[:n].numpy()
    run_b()
    xb, med, mn = timed(run_b, args.reps)
    res['B_hbm_then_copy_ms'] = {'median': med, 'min': mn}
    assert np.array_equal(xb, ref_xy)

    # C: round 1's torch ops
    hits = DeviceResult(torch, eng.device, 0, R, abi.OUT_HITS, want_pupil=False, nan_fill=False)

    def run_c():
        eng.trace_pupil_grid(f, grid, wi, o_hits, want_pupil=False, out=hits)
        idx = torch.nonzero(hits.status == 0).squeeze(1)
        xy_dev = torch.stack((hits.seg[0].index_select(0, idx), hits.seg[1].index_select(0, idx)), dim=1)
        xy_pin[:xy_dev.shape[0]].copy_(xy_dev, non_blocking=True)
        torch.cuda.synchronize()
        return xy_pin[:xy_dev.shape[0]].numpy()
    run_c()
    xc, med, mn = timed(run_c, args.reps)
    res['C_r01_torch_ops_ms'] = {'median': med, 'min': mn}
    assert np.array_equal(xc, ref_xy)

    # kernel-only durations
    res['kernel_hits_ms'] = eng.time_pupil_grid(f, grid, wi, o_hits, hits, 20)

    class _O:
        def __init__(self, o):
            self._o = o

        def out_struct(self):
            return self._o
    res['kernel_compact_hbm_ms'] = eng.time_pupil_grid(f, grid, wi, o_cmp, _O(out_b), 20)
    lease, out_p, _ = eng._hits_out(R)
    res['kernel_compact_pinned_ms'] = eng.time_pupil_grid(f, grid, wi, o_cmp, _O(out_p), 20)
    res['pcie_GBps_pinned'] = xy.shape[0] * 16 / (res['kernel_compact_pinned_ms'] * 1e-3) / 1e9
    print(json.dumps(res))


if __name__ == '__main__':
    main()

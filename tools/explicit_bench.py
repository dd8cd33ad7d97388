#!/usr/bin/env python3
"""times the explicit-ray entry (rox_trace_rays, the trace_list_of_rays shape):
rays resident in HBM, FULL packets, per-ray wavelength indices."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    num = 1024
    R = num * num
    flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES
    # ray starts = the pupil grid's, taken from a device trace (segment 0)
    o = make_opts(flags=flags | abi.APPLY_VIGNETTING, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2)
    full = eng.trace_pupil_grid(wl.fields[1], make_grid((-1., -1.), (1., 1.), num), 1, o, nan_fill=True)
    pt0 = full.seg[0, 0:3].contiguous()
    dir0 = full.seg[0, 3:6].contiguous()
    ok0 = ~torch.isnan(pt0[0])
    pt0 = torch.where(ok0, pt0, torch.zeros_like(pt0))
    dir0 = torch.where(ok0, dir0, torch.tensor([[0.], [0.], [1.]], dtype=torch.float64, device=pt0.device).expand_as(dir0))
    del full
    out = DeviceResult(torch, eng.device, N, R, abi.OUT_FULL, want_pupil=False, nan_fill=False)
    res = {}
    for name, wi in (('one_wvl', 1), ('per_ray_wvl', torch.randint(0, 3, (R,), dtype=torch.int32, device=pt0.device))):
        o2 = make_opts(flags=flags, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2)
        for _ in range(3):
            eng.trace_rays(pt0, dir0, wi, o2, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.trace_rays(pt0, dir0, wi, o2, out=out)
        e1.record()
        torch.cuda.synchronize()
        res[name + '_us'] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    res['rays'] = R
    print(json.dumps(res))


if __name__ == '__main__':
    main()

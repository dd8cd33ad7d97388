#!/usr/bin/env python3
"""The reference publishes one timing table for this path
(rayoptics/raytr/tests/trace_results.txt: `rt.trace` repeated on one ray, best
of 5).  This prints the device figures for the same kind
of models: 2^20-ray pupil grids per model, FULL packets and HITS, kernel time
from HIP events."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# rays/s of the reference's own table, by model (trace_results.txt lines 1,4,9,10)
REFERENCE_RAYS_PER_S = {'singlet_c1': 7955, 'dblgauss_c2': 2282, 'rc_telescope_c4': 7931,
                        'cell_phone': 665, 'nikkor_c3': None}


def main():
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    num = 1024
    R = num * num
    rows = []
    for name in ('singlet_c1', 'dblgauss_c2', 'rc_telescope_c4', 'cell_phone', 'nikkor_c3', 'zmx_evenasph_c3'):
        wl = workloads.load(name)
        N = wl.n_ifcs
        eng = TraceEngine(wl.table)
        fld = wl.fields[0]
        wi = wl.ref_wvl_idx
        flags = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
        grid = make_grid((-1., -1.), (1., 1.), num)
        rec = {'model': name, 'interfaces': N}
        for mode, key in ((abi.OUT_FULL, 'full'), (abi.OUT_HITS, 'hits')):
            o = make_opts(flags=flags, out_mode=mode, first_surf=1, last_surf=N - 2,
                          foc=wl.foc, image_pt=wl.image_pts[0])
            out = DeviceResult(torch, eng.device, eng.num_segments(flags), R, mode,
                               want_pupil=False, nan_fill=False)
            ms = eng.time_pupil_grid_sustained(fld, grid, wi, o, out)     # steady-state clocks
            st = out.status.cpu().numpy()
            fs = out.fail_surf.cpu().numpy().astype('int64')
            ok = st == 0
            inters = int(ok.sum()) * (N - 1) + int(fs[~ok].sum())
            rec[key + '_us'] = round(ms * 1e3, 1)
            rec[key + '_rays_per_s'] = R / (ms * 1e-3)
            rec[key + '_intersections_per_s'] = inters / (ms * 1e-3)
            rec['rays_through'] = int(ok.sum())
        ref = REFERENCE_RAYS_PER_S.get(name)
        rec['reference_rays_per_s_published'] = ref
        if ref:
            rec['full_speedup_vs_published'] = rec['full_rays_per_s'] / ref
        rows.append(rec)
        eng.close()
    print(json.dumps(rows, indent=1))


if __name__ == '__main__':
    main()

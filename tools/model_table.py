#!/usr/bin/env python3
"""The reference publishes one timing table for this path
(rayoptics/raytr/tests/time_trace.py -> trace_results.txt: `rt.trace` repeated on ONE ray --
pupil (0.5, 0.5) of field 1 at the central wavelength -- best of 5).
This prints the device figures for the same ten models (plus the round's other fixtures):

  * 2^20-ray pupil grids of that field, FULL packets and HITS, steady-state kernel time from
    HIP events (the batch form the reference does not have);
  * the benchmark's own ray through the one-ray entry (engine.trace_one: the core of the
    rebound raytrace.trace seam), call -> packet on the host.

    python tools/model_table.py > profiles/r03_models.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (workload, row of rayoptics/raytr/tests/trace_results.txt, trials, best time in s)
MODELS = [('tt_singlet_seq', 'singlet', 100000, 12.57),
          ('tt_landscape', 'landscape lens', 80000, 12.91),
          ('tt_triplet', 'Sasian triplet', 50000, 13.37),
          ('dblgauss_c2', 'double gauss', 30000, 13.15),
          ('tt_two_sph_mirrors', '2 spherical mirrors (spheres)', 100000, 11.98),
          ('tt_two_mirrors_conic', '2 spherical mirrors (conics)', 100000, 12.51),
          ('tt_paraboloid', 'paraboloid', 100000, 12.57),
          ('tt_cassegrain', 'Cassegrain', 100000, 12.54),
          ('rc_telescope_c4', 'Ritchey-Chretien', 100000, 12.61),
          ('cell_phone', 'cell phone camera', 10000, 15.05),
          ('singlet_c1', None, None, None), ('nikkor_c3', None, None, None),
          ('zmx_evenasph_c3', None, None, None)]


def main():
    import numpy as np
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid, DeviceResult
    num = 1024
    R = num * num
    rows = []
    for name, ref_row, trials, best in MODELS:
        wl = workloads.load(name)
        N = wl.n_ifcs
        eng = TraceEngine(wl.table)
        fi = min(1, len(wl.fields) - 1) if ref_row else 0       # lookup_fld_wvl_focus(1)
        fld = wl.fields[fi]
        wi = wl.ref_wvl_idx
        flags = abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
        if not (fld.kind == abi.FLD_EPD_WIDE or fld.z_dir0 == 0.0):
            flags |= abi.INTERSECT_OBJ
        grid = make_grid((-1., -1.), (1., 1.), num)
        rec = {'model': name, 'reference_benchmark_row': ref_row, 'interfaces': N, 'field': fi}
        for mode, key in ((abi.OUT_FULL, 'full'), (abi.OUT_HITS, 'hits')):
            o = make_opts(flags=flags, out_mode=mode, first_surf=1, last_surf=N - 2,
                          foc=wl.foc, image_pt=wl.image_pts[fi])
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
            del out
        # the benchmark's own ray: pupil (0.5, 0.5), no aperture checks (rt.trace defaults)
        o1 = make_opts(flags=flags & ~abi.CHECK_APERTURES, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2)
        one = eng.trace_pupil_list(fld, np.array([0.5]), np.array([0.5]), wi, o1, nan_fill=True).to_host()
        if one.status[0] == 0 and (flags & abi.INTERSECT_OBJ):
            # its start point and direction: the object-surface segment of the packet
            pt0, d0 = one.seg[0, 0:3, 0].copy(), one.seg[0, 3:6, 0].copy()
            o2 = make_opts(flags=abi.INTERSECT_OBJ, out_mode=abi.OUT_FULL, first_surf=1, last_surf=N - 2)
            for _ in range(20):
                eng.trace_one(pt0, d0, wi, o2)
            ts = []
            for _ in range(300):
                t0 = time.perf_counter()
                eng.trace_one(pt0, d0, wi, o2)
                ts.append(time.perf_counter() - t0)
            rec['one_ray_call_us_median'] = round(float(np.median(ts)) * 1e6, 1)
        if ref_row:
            pub = trials / best
            rec['reference_published'] = {'trials': trials, 'best_s': best, 'rays_per_s': round(pub),
                                          'us_per_ray': round(1e6 / pub, 1), 'hardware': 'unknown (2018)'}
            rec['full_rays_per_s_over_published'] = rec['full_rays_per_s'] / pub
        rows.append(rec)
        eng.close()
    print(json.dumps(rows, indent=1))


if __name__ == '__main__':
    main()

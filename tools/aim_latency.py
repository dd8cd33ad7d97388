#!/usr/bin/env python3
"""Latency of the batched chief-ray aiming entry (rox_aim_chief_rays: every field of a model
in one launch, secant iteration per lane).  The reference aims field by field with ~10 Python
single-ray traces each on every update_optical_properties().

    python tools/aim_latency.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine
    for name in ('dblgauss_c2', 'nikkor_c3', 'cell_phone', 'rc_telescope_c4'):
        wl = workloads.load(name)
        if not wl.aim:
            continue
        eng = TraceEngine(wl.table)
        probs = []
        for m in wl.aim:
            a = abi.Aim()
            for i in range(3):
                a.pt0[i] = m['pt0'][i]
            a.z_enp, a.y_target, a.z_dir0 = m['z_enp'], 0.0, m['z_dir0']
            a.wvl_idx, a.surf, a.flip = m['wvl_idx'], m['surf'], 1
            probs.append(a)
        n_1d = len(probs)
        for m in wl.aim2d or []:            # fields off the y axis: the fsolve (hybrd) branch
            a = abi.Aim()
            for i in range(3):
                a.pt0[i] = m['pt0'][i]
            a.z_enp, a.y_target, a.z_dir0 = m['z_enp'], 0.0, m['z_dir0']
            a.wvl_idx, a.surf, a.flip = m['wvl_idx'], m['surf'], 1
            a.two_d, a.epsfcn = 1, m['epsfcn']
            probs.append(a)
        for _ in range(5):
            eng.aim_chief_rays(probs)
        t = []
        for _ in range(60):
            t0 = time.perf_counter()
            eng.aim_chief_rays(probs)
            t.append(time.perf_counter() - t0)
        print(json.dumps({'workload': name, 'interfaces': wl.n_ifcs, 'fields_aimed': len(probs),
                          'on_the_y_axis': n_1d, 'off_axis_2d': len(probs) - n_1d,
                          'ms_median': float(np.median(t) * 1e3), 'ms_min': float(np.min(t) * 1e3)}))
        eng.close()


def wide_angle():
    """rox_find_real_enp on the stored wide-angle problems (tests/golden/wideangle.npz): all
    of a model's problems in one call, and nine of them (a typical field list)"""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_oracle_wideangle import golden_model
    from rayoptics_amd import abi
    from rayoptics_amd.engine import TraceEngine
    for name in ('dblgauss', 'nikkor'):
        tbl, probs, _z, raised = golden_model(name)
        eng = TraceEngine(tbl)
        for n in (9, len(probs)):
            sel = [p for p, r in zip(probs, raised) if not r][:n]
            for _ in range(5):
                eng.find_real_enp(sel)
            t = []
            for _ in range(40):
                t0 = time.perf_counter()
                _zz, res = eng.find_real_enp(sel)
                t.append(time.perf_counter() - t0)
            print(json.dumps({'wide_angle_search': name, 'interfaces': tbl.n_ifcs, 'problems': len(sel),
                              'found': int((res == abi.ENP_FOUND).sum()),
                              'ms_median': float(np.median(t) * 1e3), 'ms_min': float(np.min(t) * 1e3)}))
        eng.close()


if __name__ == '__main__':
    if '--wide-angle' in sys.argv:
        wide_angle()
        sys.exit(0)
    main()

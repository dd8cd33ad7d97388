#!/usr/bin/env python3
"""Latency of wideangle.eval_real_image_ht (FieldSpec.obj_coords of fields given as real image
heights: BASELINE configs[2]'s .zmx import) -- the reference's own reverse chief-ray loop vs
the rebound trace.iterate_ray_raw (one launch for the iteration + one for the last trial ray)
-- and of vigcalc.set_pupil's iterate_pupil_ray.  Needs the live reference next to the GPU
(oracle/_ref, oracle/stage_reference.py).

    python tools/real_image_ht_latency.py > profiles/r04_real_image_ht_latency.jsonl"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import numpy as np  # noqa: E402


def med(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def main():
    import refmodels as ref
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import install
    import rayoptics.raytr.wideangle as wa
    import rayoptics.raytr.vigcalc as vc
    import rayoptics.raytr.raytrace as rraytrace
    opm = ref.zmx_evenasph_c3()
    osp = opm['osp']
    wvl = osp['wvls'].central_wvl
    flds = list(osp['fov'].fields)
    flds[1].x = 0.4                      # one field off the y axis: MINPACK's 2-D branch
    n_seam = {'n': 0}
    t0_, r0_ = rraytrace.trace, rraytrace.trace_raw

    def ct(*a, **k):
        n_seam['n'] += 1
        return t0_(*a, **k)

    def cr(*a, **k):
        n_seam['n'] += 1
        return r0_(*a, **k)
    rraytrace.trace, rraytrace.trace_raw = ct, cr
    for k, fld in enumerate(flds):
        n_seam['n'] = 0
        theirs = wa.eval_real_image_ht(opm, fld, wvl)
        rays = n_seam['n']
        ms_ref = med(lambda: wa.eval_real_image_ht(opm, fld, wvl), 15)
        rraytrace.trace, rraytrace.trace_raw = t0_, r0_
        install.install()
        ours = wa.eval_real_image_ht(opm, fld, wvl)
        ms_dev = med(lambda: wa.eval_real_image_ht(opm, fld, wvl), 40)
        install.uninstall()
        rraytrace.trace, rraytrace.trace_raw = ct, cr
        same = (np.array_equal(ours[0][0], theirs[0][0]) and np.array_equal(ours[0][1], theirs[0][1])
                and ours[1] == theirs[1])
        print(json.dumps({'what': 'wideangle.eval_real_image_ht, one field', 'model': 'zmx_evenasph_c3',
                          'field': k, 'branch': '2-D (hybrd)' if fld.x != 0.0 else '1-D (secant)',
                          'reference_ms': ms_ref, 'reference_single_ray_traces': rays,
                          'drop_in_ms': ms_dev, 'bit_identical': bool(same)}), flush=True)
    rraytrace.trace, rraytrace.trace_raw = t0_, r0_
    # vigcalc.iterate_pupil_ray as set_pupil calls it
    for name in ('dblgauss', 'nikkor'):
        opm = getattr(ref, name)()
        sm = opm['seq_model']
        fld0, cwl, _foc = opm['osp'].lookup_fld_wvl_focus(0)
        r_stop = sm.ifcs[sm.stop_surface].surface_od() * 0.8

        def call():
            return vc.iterate_pupil_ray(opm, sm.stop_surface, 1, 1.0, r_stop, fld0, cwl)
        theirs = call()
        ms_ref = med(call, 15)
        install.install()
        ours = call()
        ms_dev = med(call, 40)
        install.uninstall()
        print(json.dumps({'what': 'vigcalc.iterate_pupil_ray (set_pupil: marginal ray to 0.8 x the stop edge)',
                          'model': name, 'reference_ms': ms_ref, 'drop_in_ms': ms_dev,
                          'max_abs_diff': float(np.max(np.abs(ours - theirs)))}), flush=True)


if __name__ == '__main__':
    main()

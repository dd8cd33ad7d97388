#!/usr/bin/env python3
"""One-off soaks against the LIVE reference (build container only; the oracle stands in for the
device): the suite's randomized tests over many more seeds, and the aiming / vignetting / OPD /
fan drop-ins on randomly perturbed models.  Prints one JSON line per soak.

    python tools/soak_reference.py"""
import json
import logging
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
logging.disable(logging.CRITICAL)


def soak_tests():
    import test_oracle_fuzz_reference as t
    import test_oracle_phase_reference as tp
    t0, bad, n = time.time(), [], 0
    for seed in range(30, 1530):
        try:
            t.test_oracle_equals_reference_on_random_paths(seed)
            n += 1
        except AssertionError as e:
            bad.append((seed, str(e)[:120]))
    print(json.dumps({'soak': 'oracle == reference, random paths', 'systems': n, 'mismatches': bad[:5],
                      'seconds': round(time.time() - t0, 1)}))
    t0, bad, n = time.time(), [], 0
    for kind in ('grating', 'doe', 'hologram', 'thinlens'):
        for seed in range(8, 308):
            try:
                tp.test_phase_elements_oracle_equals_reference(kind, seed)
                n += 1
            except AssertionError as e:
                if str(e):                      # (an empty message = too few rays got through)
                    bad.append((kind, seed, str(e)[:120]))
    print(json.dumps({'soak': 'oracle == reference, phase elements', 'systems': n,
                      'mismatches': bad[:5], 'seconds': round(time.time() - t0, 1)}))


def perturbed(build, rng, dcv, dfov=0.0):
    opm = build()
    sm, osp = opm['seq_model'], opm['osp']
    for ifc in sm.ifcs[1:-1]:
        if hasattr(ifc, 'profile') and ifc.profile.cv != 0:
            ifc.profile.cv *= 1 + rng.uniform(-dcv, dcv)
    if dfov:
        g = sm.gaps[int(rng.integers(1, len(sm.gaps) - 1))]
        g.thi *= 1 + rng.uniform(-0.03, 0.03)
        osp['fov'].value *= 1 + rng.uniform(-dfov, 0.1)
    sm.update_model()
    osp.update_model()
    opm.update_optical_properties()
    return opm


def soak_dropins():
    import refmodels as ref
    from rayoptics_amd import session, install
    from oracle_engine import OracleEngine
    import rayoptics.raytr.vigcalc as vigcalc
    import rayoptics.raytr.analyses as analyses
    session.ENGINE_FACTORY = OracleEngine
    rng = np.random.default_rng(77)
    t0, bad, n = time.time(), [], 0
    for build in (ref.dblgauss, ref.singlet, ref.rc_telescope, ref.cell_phone, ref.nikkor):
        for trial in range(30):
            install.uninstall()
            try:
                opm = perturbed(build, rng, 0.02, 0.15)
            except Exception:
                continue
            flds = opm['osp']['fov'].fields

            def run():
                for f in flds:
                    f.aim_info = None
                opm['osp'].update_optical_properties()
                aim = [np.array(f.aim_info, dtype=float).tolist() for f in flds]
                for f in flds:
                    f.vux = f.vlx = f.vuy = f.vly = 0.0
                vigcalc.set_vig(opm)
                return aim, [(f.vux, f.vlx, f.vuy, f.vly) for f in flds]
            theirs = run()
            install.install()
            ours = run()
            install.uninstall()
            n += len(flds)
            if ours != theirs:
                bad.append((build.__name__, trial))
    print(json.dumps({'soak': 'aim_info and vignetting factors on perturbed models', 'fields': n,
                      'mismatches': bad[:5], 'seconds': round(time.time() - t0, 1)}))
    rng = np.random.default_rng(91)
    t0, bad, n = time.time(), [], 0
    for build in (ref.dblgauss, ref.telecentric, ref.rc_telescope, ref.cell_phone, ref.singlet):
        for trial in range(25):
            install.uninstall()
            try:
                opm = perturbed(build, rng, 0.01)
            except Exception:
                continue
            osp = opm['osp']
            fld = osp['fov'].fields[int(rng.integers(0, len(osp['fov'].fields)))]
            wvl = osp['wvls'].wavelengths[int(rng.integers(0, len(osp['wvls'].wavelengths)))]
            foc = float(rng.uniform(-0.05, 0.05))

            def run():
                a = np.array(analyses.eval_wavefront(opm, fld, wvl, foc, num_rays=7), dtype=float)
                gp = analyses.trace_wavefront(opm, fld, wvl, foc, num_rays=7)
                b = np.array(analyses.focus_wavefront(opm, gp, fld, wvl, foc + 0.01), dtype=float)
                fan = [np.r_[np.ravel(x[0]), np.ravel(x[1])].tolist() if len(x) == 2 else list(x)
                       for x in analyses.eval_fan(opm, fld, wvl, foc, 1, num_rays=9)]
                return a, b, fan
            try:
                theirs = run()
            except Exception:
                continue
            install.install()
            ours = run()
            install.uninstall()
            n += 1
            same = (np.array_equal(ours[0], theirs[0], equal_nan=True)
                    and np.array_equal(ours[1], theirs[1], equal_nan=True)
                    and json.dumps(ours[2]) == json.dumps(theirs[2]))
            if not same:
                bad.append((build.__name__, trial))
    print(json.dumps({'soak': 'eval_wavefront / trace+focus_wavefront / eval_fan on perturbed models',
                      'cases': n, 'mismatches': bad[:5], 'seconds': round(time.time() - t0, 1)}))
    session.ENGINE_FACTORY = None


if __name__ == '__main__':
    soak_tests()
    soak_dropins()

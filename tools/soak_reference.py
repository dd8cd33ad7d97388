#!/usr/bin/env python3
"""One-off soaks against the LIVE reference (build container only; the oracle stands in for the
device): the suite's randomized tests over many more seeds, and the aiming / vignetting / OPD /
fan drop-ins on randomly perturbed models.  Prints one JSON line per soak.

    python tools/soak_reference.py [--round3 | --round4]
    python tools/soak_reference.py --hip        (GPU box: the HIP engine instead of the oracle)"""
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


def soak_dropins(hip=False):
    """hip=True (GPU box, reference staged in oracle/_ref): the same soaks with the HIP engine
    behind the drop-ins instead of the oracle double -- HIP vs the live reference directly;
    the vignetting factors then agree to 1e-11 instead of bit for bit (libm pow)"""
    import refmodels as ref
    from rayoptics_amd import session, install
    import rayoptics.raytr.vigcalc as vigcalc
    import rayoptics.raytr.analyses as analyses
    if hip:
        session._set_engine_factory(None)
    else:
        from oracle_engine import OracleEngine
        session._set_engine_factory(OracleEngine)
    eng = 'HIP engine' if hip else 'oracle double'
    rng = np.random.default_rng(177 if hip else 77)
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
            if hip:
                same = ours[0] == theirs[0] and np.allclose(np.array(ours[1], dtype=float),
                                                            np.array(theirs[1], dtype=float), rtol=0, atol=1e-11)
            else:
                same = ours == theirs
            if not same:
                bad.append((build.__name__, trial))
    print(json.dumps({'soak': 'aim_info and vignetting factors on perturbed models', 'engine': eng, 'fields': n,
                      'mismatches': bad[:5], 'seconds': round(time.time() - t0, 1)}))
    rng = np.random.default_rng(191 if hip else 91)
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
                      'engine': eng, 'cases': n, 'mismatches': bad[:5], 'seconds': round(time.time() - t0, 1)}))
    if hip:
        soak_packets_hip(ref, install)
    session._set_engine_factory(None)


def soak_packets_hip(ref, install):
    """trace.trace_grid packets and SpotDiagramFigure data of perturbed models: HIP engine
    behind the drop-ins vs the reference's per-ray loop, bit for bit"""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    import rayoptics.raytr.trace as rtrace
    from rayoptics.mpl.axisarrayfigure import SpotDiagramFigure
    rng = np.random.default_rng(4242)
    t0, n_models, n_rays, n_fig, bad = time.time(), 0, 0, 0, []
    for build in (ref.dblgauss, ref.nikkor, ref.cell_phone, ref.rc_telescope, ref.tilted_singlet,
                  ref.zmx_evenasph_c3):
        for trial in range(6):
            install.uninstall()
            try:
                opm = perturbed(build, rng, 0.01) if build is not ref.zmx_evenasph_c3 else build()
            except Exception:
                continue
            osp = opm['osp']
            fld = osp['fov'].fields[int(rng.integers(0, len(osp['fov'].fields)))]
            wvl = osp['wvls'].wavelengths[int(rng.integers(0, len(osp['wvls'].wavelengths)))]

            def packets():
                got = []
                rtrace.trace_grid(opm, [np.array([-1., -1.]), np.array([1., 1.]), 20], fld, wvl, 0.0,
                                  img_filter=lambda p, pkg: got.append((np.array(p), pkg)), form='list',
                                  append_if_none=True)
                out = []
                for p, pkg in got:
                    if pkg is None:
                        out.append((p.tolist(), None))
                    else:
                        ray, op, w = pkg
                        out.append((p.tolist(), [[np.asarray(sg[0]).tolist(), np.asarray(sg[1]).tolist(),
                                                  float(sg[2]), np.asarray(sg[3]).tolist()] for sg in ray],
                                    float(op), float(w)))
                return out

            def figure():
                fig = plt.figure(FigureClass=SpotDiagramFigure, opt_model=opm, num_rays=12)
                fig.update_data()
                d = [[np.array(g).tolist() for g in row[0][0]] for row in fig.axis_data_array]
                plt.close(fig)
                return d
            try:
                theirs = (packets(), figure())
            except Exception:
                continue
            install.install()
            ours = (packets(), figure())
            install.uninstall()
            n_models += 1
            n_rays += len(theirs[0])
            n_fig += 1
            # (== on the nested lists: -0.0 equals 0.0.  The reference's double Gauss data gives
            # its flat surfaces the *integer* curvature 0, and `-0 * x` is +0 where `-0.0 * x` is
            # -0: zero components of its normals carry the other sign than a float table's)
            if ours != theirs:
                bad.append((build.__name__, trial))
    print(json.dumps({'soak': 'trace.trace_grid packets (20 x 20, every ray: segments, op_delta, None pattern) '
                              'and SpotDiagramFigure data on perturbed models', 'engine': 'HIP engine',
                      'models': n_models, 'rays': n_rays, 'figures': n_fig, 'mismatches': bad[:5],
                      'n_mismatches': len(bad), 'seconds': round(time.time() - t0, 1)}))


def soak_round3(hip=False):
    """(hip=True, GPU box: the drop-ins over the HIP engine instead of the oracle restatement)
    round 3: the wide-angle pupil search restated in the oracle against the reference's
    find_real_enp on random field angles of perturbed models (z_enp bit for bit; "the reference
    raises" exactly where it does), 2-D chief-ray aiming (fsolve) through the drop-in on random
    off-axis fields, SequentialModel.trace_fan with RayFanFigure's callbacks through the fused
    batch"""
    import warnings
    import refmodels as ref
    import rayoptics.raytr.wideangle as wa
    import rayoptics.raytr.trace as rtrace
    from rayoptics_amd import abi, session, install, SurfaceTable
    from rayoptics_amd import trace as T
    from oracle import oracle
    from oracle_engine import OracleEngine
    warnings.simplefilter('ignore')
    rng = np.random.default_rng(303)
    nik = os.path.join(ref.REF_SRC, 'rayoptics', 'optical', 'tests', 'Nikon Nikkor Z 14-30mm f-4 S.roa')
    builders = [('dblgauss', ref.dblgauss, 75.),
                ('nikkor', lambda: ref.load_roa(nik, fov=(('object', 'angle'), 57.7), flds=[0., 30., 57.7],
                                                 is_relative=False), 80.)]
    t0, n, bad, codes = time.time(), 0, [], {}
    for name, build, top in builders:
        for trial in range(12):
            opm = build()
            sm = opm['seq_model']
            if trial:
                for ifc in sm.ifcs[1:-1]:
                    if hasattr(ifc, 'profile') and ifc.profile.cv != 0:
                        ifc.profile.cv *= 1 + 0.1 * rng.normal()
                ref.finish(opm, do_apertures=False)
            fov = opm['osp']['fov']
            fov.is_wide_angle = True
            tbl = SurfaceTable.from_seq_model(sm)
            for ang in rng.uniform(0., top, 40):
                fld = fov.fields[-1]
                fld.x, fld.y, fld.aim_info = 0., float(ang) / (fov.value if fov.is_relative else 1.0), None
                wvl = sm.central_wavelength()
                if hip:
                    # the rebound wideangle.find_real_enp (the HIP launch; where the reference
                    # raises it calls the reference's own function) against the reference
                    def call():
                        fld.aim_info = None
                        try:
                            return float(wa.find_real_enp(opm, sm.stop_surface, fld, wvl)[0])
                        except Exception as e:
                            return type(e).__name__
                    install.uninstall()
                    z_ref = call()
                    session._set_engine_factory(None)
                    install.install()
                    z_dev = call()
                    install.uninstall()
                    ok = z_dev == z_ref
                    codes[type(z_ref).__name__] = codes.get(type(z_ref).__name__, 0) + 1
                    n += 1
                    if not ok:
                        bad.append((name, trial, float(ang), str(z_dev), str(z_ref)))
                    continue
                pb = T._enp_problem(opm, fld, wvl, tbl, sm.stop_surface)
                z, res = oracle.find_real_enp(tbl, [pb])
                codes[int(res[0])] = codes.get(int(res[0]), 0) + 1
                try:
                    z_ref, _rr = wa.find_real_enp(opm, sm.stop_surface, fld, wvl)
                    ok = res[0] != abi.ENP_REFERENCE_RAISES and z[0, 0] == float(z_ref)
                except Exception:
                    ok = res[0] == abi.ENP_REFERENCE_RAISES
                n += 1
                if not ok:
                    bad.append((name, trial, float(ang)))
    print(json.dumps({'soak': 'wide-angle pupil search: ' + ('drop-in over the HIP engine' if hip else 'oracle')
                              + ' == reference find_real_enp', 'cases': n,
                      'mismatches': bad[:5], 'result_codes': {str(k): v for k, v in sorted(codes.items())},
                      'seconds': round(time.time() - t0, 1)}))
    # 2-D aiming and the fan figure callbacks through the drop-ins (oracle as the engine)
    session._set_engine_factory(None if hip else OracleEngine)
    eng = 'HIP engine' if hip else 'oracle double'
    t0, n, bad = time.time(), 0, []
    for build in (ref.dblgauss, ref.nikkor, ref.cell_phone):
        for trial in range(25):
            opm = perturbed(build, rng, 0.01)
            fld = opm['osp']['fov'].fields[-1]
            fld.x, fld.y = float(rng.uniform(-0.7, 0.7)), float(rng.uniform(-0.9, 0.9))

            def run():
                fld.aim_info = None
                return np.array(rtrace.aim_chief_ray(opm, fld), dtype=float)
            try:
                theirs = run()
            except Exception:
                continue
            install.install()
            ours = run()
            install.uninstall()
            n += 1
            if not np.array_equal(ours, theirs):
                bad.append((build.__name__, trial))
    print(json.dumps({'soak': '2-D chief-ray aiming (fsolve / hybrd) through the drop-in on perturbed models', 'engine': eng,
                      'cases': n, 'mismatches': bad[:5], 'seconds': round(time.time() - t0, 1)}))
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    from rayoptics.mpl.axisarrayfigure import RayFanFigure
    t0, n, bad = time.time(), 0, []
    for build in (ref.dblgauss, ref.nikkor, ref.telecentric, ref.rc_telescope):
        for trial in range(6):
            try:
                opm = perturbed(build, rng, 0.005)
            except Exception:
                continue
            for data_type in ('Ray', 'OPD'):
                def run():
                    fig = plt.figure(FigureClass=RayFanFigure, opt_model=opm, data_type=data_type,
                                     do_smoothing=False, num_rays=11)
                    fig.update_data()
                    d = [[(np.array(c[0]).tolist(), np.array(c[1]).tolist(), c[2]) for c in row]
                         for row in fig.axis_data_array]
                    plt.close(fig)
                    return d
                try:
                    theirs = run()
                except Exception:
                    continue
                install.install()
                ours = run()
                install.uninstall()
                n += 1
                if json.dumps(ours) != json.dumps(theirs):
                    bad.append((build.__name__, trial, data_type))
    print(json.dumps({'soak': 'RayFanFigure (Ray / OPD) through SequentialModel.trace_fan on perturbed models', 'engine': eng,
                      'figures': n, 'mismatches': bad[:5], 'seconds': round(time.time() - t0, 1)}))
    session._set_engine_factory(None)


def soak_round4(hip=False):
    """round 4: the reverse chief-ray iteration (trace.iterate_ray_raw behind
    wideangle.eval_real_image_ht) through the drop-in, oracle as the engine, against the
    reference's own loop on perturbed copies of the .zmx import (real-image-height fields on
    and off the y axis: scipy's secant iteration and MINPACK's hybrd)"""
    import time
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import refmodels as ref
    from rayoptics_amd import session, install
    from oracle_engine import OracleEngine
    import rayoptics.raytr.wideangle as wa
    rng = np.random.default_rng(144 if hip else 44)
    session._set_engine_factory(None if hip else OracleEngine)
    t0, n, n2d, n_exc, bad = time.time(), 0, 0, 0, []
    for trial in range(120):
        opm = ref.zmx_evenasph_c3()
        sm = opm['seq_model']
        scale = float(rng.choice([1e-3, 5e-3, 2e-2, 5e-2]))
        for ifc in sm.ifcs[1:-1]:
            ifc.profile.cv *= 1.0 + scale * rng.normal()
        for g in sm.gaps[1:-1]:
            g.thi *= 1.0 + 0.3 * scale * rng.normal()
        try:
            ref.finish(opm, do_apertures=False)
        except Exception:
            continue
        osp = opm['osp']
        wvl = osp['wvls'].central_wvl
        for fld in osp['fov'].fields:
            fld.y *= float(rng.uniform(0.2, 1.3))
            if rng.random() < 0.5:
                fld.x = float(rng.normal()) * 0.5

            def run():
                try:
                    (p_o, d_o), z = wa.eval_real_image_ht(opm, fld, wvl)
                    return [np.array(p_o).tolist(), np.array(d_o).tolist(), float(z)]
                except Exception as e:
                    return type(e).__name__
            theirs = run()
            install.install()
            ours = run()
            install.uninstall()
            n += 1
            n2d += fld.x != 0.0
            n_exc += isinstance(theirs, str)
            if json.dumps(ours) != json.dumps(theirs):
                bad.append((trial, float(fld.x), float(fld.y), str(ours)[:80], str(theirs)[:80]))
    print(json.dumps({'soak': 'wideangle.eval_real_image_ht through the rebound trace.iterate_ray_raw '
                              '(' + ('HIP engine' if hip else 'oracle as the engine') + ') == the reference, bit for bit',
                      'cases': n, 'two_d_cases': int(n2d), 'cases_where_both_raise': int(n_exc),
                      'mismatches': bad[:5], 'n_mismatches': len(bad), 'seconds': round(time.time() - t0, 1)}))
    session._set_engine_factory(None)


if __name__ == '__main__':
    if '--hip' in sys.argv:     # GPU box, reference staged in oracle/_ref: HIP vs the live reference
        soak_dropins(hip=True)
        soak_round4(hip=True)
        soak_round3(hip=True)
        sys.exit(0)
    if '--round4' in sys.argv:
        soak_round4()
        sys.exit(0)
    if '--round3' in sys.argv:
        soak_round3()
        sys.exit(0)
    soak_tests()
    soak_dropins()

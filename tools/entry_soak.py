#!/usr/bin/env python3
"""Soak of the small entries, device vs oracle: every ray-start branch with random field
constants (bit for bit), chief-ray aiming on perturbed problems (bit for bit), the vignetting
search on perturbed fields (<= 1e-12: the objective squares a coordinate, libm pow in the
reference and the oracle, an exact product in the kernel), the wide-angle pupil search on
random directions / pupil guesses / starting points (bit for bit, result codes included).

    python tools/entry_soak.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine
    from rayoptics_amd.table import field_struct
    from oracle import oracle
    import helpers as H
    out = {}
    t0 = time.time()
    # ---- ray starts
    fx = H.fixture('singlet')
    eng = TraceEngine(fx.table)
    N = fx.table.n_ifcs
    z0 = float(fx.table.rows[0].t[2])
    rng = np.random.default_rng(5)
    n_bad = n = 0
    for kind in (abi.FLD_EPD, abi.FLD_EPD_WIDE, abi.FLD_AIM_PT, abi.FLD_NA, abi.FLD_FNO, abi.FLD_AIM_DIR):
        for trial in range(150):
            ang = np.deg2rad(rng.uniform(-12, 12))
            rot = np.array([[1, 0, 0], [0, np.cos(ang), -np.sin(ang)], [0, np.sin(ang), np.cos(ang)]])
            if trial % 2:
                rot = np.asfortranarray(rot)
            scale = {abi.FLD_EPD: rng.uniform(1, 6), abi.FLD_EPD_WIDE: rng.uniform(1, 6), abi.FLD_AIM_PT: 0.0,
                     abi.FLD_NA: rng.uniform(0.01, 0.08), abi.FLD_FNO: -1 / rng.uniform(6, 20),
                     abi.FLD_AIM_DIR: 0.0}[kind]
            fld = field_struct((rng.uniform(-1, 1) if trial % 3 == 0 else 0.0, rng.uniform(-3, 3), 0.0),
                               (rng.uniform(-.1, .1), rng.uniform(-.3, .3)), scale, z0 + rng.uniform(-2, 2),
                               tuple(rng.uniform(0, 0.3, 4)), 1.0, kind=kind, rot=rot,
                               cr_dir=(rng.uniform(-.02, .02), rng.uniform(-.03, .03)))
            span = {abi.FLD_AIM_PT: 4.0, abi.FLD_AIM_DIR: 0.05}.get(kind, 1.0)
            grid = oracle.make_grid((-span, -span), (span, span), 17)
            flags = abi.CHECK_APERTURES | abi.APPLY_VIGNETTING | (0 if kind == abi.FLD_EPD_WIDE else abi.INTERSECT_OBJ)
            opts = oracle.make_opts(flags=flags, first_surf=1, last_surf=N - 2)
            with np.errstate(all='ignore'):
                orc = oracle.trace_pupil_grid(fx.table, fld, grid, 0, opts)
            dev = eng.trace_pupil_grid(fld, grid, 0, opts, nan_fill=True).to_host()
            same = (np.array_equal(dev.status, orc.status)
                    and np.array_equal(dev.seg, orc.seg, equal_nan=True)
                    and np.array_equal(dev.op, orc.op, equal_nan=True)
                    and np.array_equal(dev.pupil, orc.pupil, equal_nan=True))
            n += 1
            n_bad += not same
    eng.close()
    out['ray_starts'] = {'fields': n, 'mismatching': n_bad}
    # ---- aiming and vignetting on perturbed problems
    n_aim = bad_aim = n_vig = bad_clip = n_lane = bad_lane = 0
    worst_vig = 0.0
    for name in ('dblgauss_c2', 'nikkor_c3', 'cell_phone', 'singlet_c1', 'rc_telescope_c4'):
        wl = workloads.load(name)
        eng = TraceEngine(wl.table)
        probs = []
        for trial in range(60):
            for m in wl.aim or []:
                a = abi.Aim()
                f = rng.uniform(0.3, 1.3)
                a.pt0[0], a.pt0[1], a.pt0[2] = 0.0, m['pt0'][1] * f, m['pt0'][2]
                a.z_enp = m['z_enp'] * rng.uniform(0.95, 1.05)
                a.y_target, a.z_dir0 = 0.0, m['z_dir0']
                a.wvl_idx, a.surf, a.flip = int(rng.integers(0, len(wl.table.wvls))), m['surf'], 1
                probs.append(a)
            for m in wl.aim2d or []:            # the fsolve branch: fields off the y axis
                a = abi.Aim()
                a.pt0[0] = m['pt0'][0] * rng.uniform(0.3, 1.5)
                a.pt0[1], a.pt0[2] = m['pt0'][1] * rng.uniform(0.3, 1.5), m['pt0'][2]
                a.z_enp = m['z_enp'] * rng.uniform(0.95, 1.05)
                a.x_target = a.y_target = 0.0
                a.z_dir0 = m['z_dir0']
                a.wvl_idx, a.surf, a.flip = int(rng.integers(0, len(wl.table.wvls))), m['surf'], 1
                a.two_d, a.epsfcn = 1, m['epsfcn']
                probs.append(a)
        if probs:
            yd, rd = eng.aim_chief_rays(probs)
            yo, ro = oracle.aim_chief_rays(wl.table, probs)
            n_aim += len(probs)
            bad_aim += int((~((yd == yo) | (np.isnan(yd) & np.isnan(yo))).all(axis=1)).sum() + (rd != ro).sum())
            # up to 1024 problems run a wave each (the independent trial rays of a search traced
            # side by side in its lanes), more run a lane each: the same answers either way
            k = 1024 // len(probs) + 1
            yl, rl = eng.aim_chief_rays(probs * k)
            n_lane += len(probs) * k
            bad_lane += int((~((yl == np.tile(yo, (k, 1))) | np.isnan(yl)).all(axis=1)).sum() + (rl != np.tile(ro, k)).sum())
        vp = []
        for trial in range(20):
            for v in wl.vig or []:
                for i, start in enumerate(v['starts']):
                    p = abi.Vig()
                    p.fld = wl.fields[v['field_index']]
                    p.fld.pt0[1] *= rng.uniform(0.6, 1.1)
                    p.fld.aim[1] *= rng.uniform(0.9, 1.1)
                    s = np.array(start, dtype=float)
                    u = s / np.linalg.norm(s)
                    p.start_dir[0], p.start_dir[1] = s
                    p.unit_dir[0], p.unit_dir[1] = u
                    p.xy, p.wvl_idx, p.stop_surf, p.max_iter = i // 2, v['wvl_idx'], v['stop'], 50
                    vp.append(p)
        if vp:
            vd, cd = eng.calc_vignetting(vp)
            vo, co = oracle.calc_vignetting(wl.table, vp)
            n_vig += len(vp)
            bad_clip += int((cd != co).sum())
            k = 1024 // len(vp) + 1
            vl, cl = eng.calc_vignetting(vp * k)
            n_lane += len(vp) * k
            bad_lane += int((cl != np.tile(cd, k)).sum() + (~((vl == np.tile(vd, k)) | np.isnan(vl))).sum())
            ok = np.isfinite(vd) & np.isfinite(vo)
            worst_vig = max(worst_vig, float(np.abs(vd[ok] - vo[ok]).max()) if ok.any() else 0.0)
        eng.close()
    # ---- the wide-angle pupil search: random field directions, pupil guesses and starting
    # points on three tables (problems need no reference: any direction / rotation pair is a
    # valid input to the search)
    n_enp = bad_enp = 0
    codes = {}
    for name in ('dblgauss_c2', 'nikkor_c3', 'cell_phone'):
        wl = workloads.load(name)
        eng = TraceEngine(wl.table)
        stop = wl.table.stop_idx if getattr(wl.table, 'stop_idx', None) is not None else 1
        z0 = float(wl.table.rows[0].t[2])
        probs = []
        for trial in range(600):
            ang = np.deg2rad(rng.uniform(0., 88.) if trial % 4 else rng.uniform(0., 20.))
            e = abi.Enp()
            d = np.array([0., np.sin(ang), np.cos(ang)])
            # rot_v1_into_v2([0, 0, 1], d) for a direction in the y-z plane: a rotation about x
            rot = np.array([[1., 0., 0.], [0., np.cos(ang), np.sin(ang)], [0., -np.sin(ang), np.cos(ang)]])
            for i in range(3):
                e.dir0[i] = d[i]
                for j in range(3):
                    e.rot[3 * i + j] = rot[i, j]
            e.rot_order = abi.RT_C_ORDER
            e.obj_dist = z0
            e.z_enp_0 = float(rng.uniform(2., 80.)) * (1 if trial % 7 else -1)
            e.aim_info = float('nan') if trial % 5 else e.z_enp_0 * rng.uniform(0.5, 1.5)
            e.wvl_idx = int(rng.integers(0, len(wl.table.wvls)))
            e.surf = int(stop) if trial % 11 else max(1, int(stop) - 1)
            e.check_direction = 1 if trial % 13 else 0
            probs.append(e)
        zd, rd = eng.find_real_enp(probs)
        zo, ro = oracle.find_real_enp(wl.table, probs)
        n_enp += len(probs)
        bad_enp += int(((rd != ro) | ~((zd == zo) | (np.isnan(zd) & np.isnan(zo))).all(axis=1)).sum())
        for c in ro.tolist():
            codes[c] = codes.get(c, 0) + 1
        eng.close()
    out['wide_angle_search'] = {'problems': n_enp, 'mismatching': bad_enp,
                                'oracle_result_codes': {str(k): v for k, v in sorted(codes.items())}}
    out['aiming'] = {'problems': n_aim, 'mismatching': bad_aim}
    out['vignetting'] = {'problems': n_vig, 'clip_surface_mismatches': bad_clip, 'max_abs_diff': worst_vig}
    out['lane_per_problem_vs_wave_per_problem'] = {'problems': n_lane, 'mismatching': bad_lane}
    out['seconds'] = round(time.time() - t0, 1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

"""Generates tests/golden/*.npz by RUNNING THE REFERENCE ITSELF
(mjhoptics/ray-optics at /root/reference, imported through oracle/refshim.py).

    python tests/golden/make_golden.py

Runs only in the build container.  Each fixture holds
  * the flat surface table (JSON, rayoptics_amd.SurfaceTable.to_dict) read
    from the live reference SequentialModel,
  * the inputs (explicit rays, or the per-field constants + grid definition),
  * the reference's outputs, converted to the SoA layout of include/roxtrace.h
    (seg[K][10][R], NaN where the reference produced nothing).
The outputs come from the reference's own drivers:
  rays  -> rayoptics.raytr.raytrace.trace            (raytrace.py:51-80)
  grid  -> rayoptics.raytr.trace.trace_grid          (trace.py:563-605), every
           RayResult recorded by wrapping trace.trace_safe
  fan   -> rayoptics.raytr.trace.trace_fan           (trace.py:537-560)
  spot  -> rayoptics.mpl.axisarrayfigure.SpotDiagramFigure.update_data
           (axisarrayfigure.py:222-263) -> SequentialModel.trace_grid
  list  -> rayoptics.raytr.analyses.trace_list_of_rays (analyses.py:458-510)
  psf   -> rayoptics.raytr.analyses.calc_psf          (analyses.py:848-875)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..', '..'))

import refmodels as rm  # noqa: E402  (installs the reference shim)
import rayoptics_amd as ra  # noqa: E402
from rayoptics_amd import abi  # noqa: E402
from rayoptics_amd.table import field_from_model  # noqa: E402

import rayoptics.raytr.raytrace as rt  # noqa: E402
import rayoptics.raytr.trace as trace  # noqa: E402
import rayoptics.raytr.analyses as analyses  # noqa: E402
from rayoptics.raytr import traceerror as terr  # noqa: E402
from rayoptics.seq.sequential import gen_sequence  # noqa: E402

SEED = 20260925


def status_of(err):
    if err is None:
        return abi.OK
    if isinstance(err, terr.TraceMissedSurfaceError):
        return abi.MISSED_SURFACE
    if isinstance(err, terr.TraceTIRError):
        return abi.TIR
    if isinstance(err, terr.TraceRayBlockedError):
        return abi.BLOCKED
    if isinstance(err, terr.TraceEvanescentRayError):
        return abi.EVANESCENT
    raise err


class SoA:
    """reference RayPkgs -> seg[K][10][R] + op + status + fail_surf"""

    def __init__(self, K, R):
        self.seg = np.full((K, 10, R), np.nan)
        self.op = np.full(R, np.nan)
        self.status = np.full(R, 255, dtype=np.uint8)
        self.fail_surf = np.full(R, -2, dtype=np.int16)

    def put(self, r, pkg, err):
        self.status[r] = status_of(err)
        self.fail_surf[r] = -1 if err is None else err.surf
        if pkg is None:
            return
        ray, op, _wvl = pkg[0], pkg[1], pkg[2]
        self.op[r] = op
        for k, s in enumerate(ray):
            self.seg[k, 0:3, r] = s[0]
            self.seg[k, 3:6, r] = s[1]
            self.seg[k, 6, r] = s[2]
            self.seg[k, 7:10, r] = s[3]

    def arrays(self, prefix=''):
        return {prefix + 'seg': self.seg, prefix + 'op': self.op,
                prefix + 'status': self.status,
                prefix + 'fail_surf': self.fail_surf}


def field_arr(f):
    return np.array(list(f.pt0) + list(f.aim) + [f.eprad, f.z_enp, f.vlx, f.vux,
                                                  f.vly, f.vuy, f.z_dir0])


def trace_with_errors(sm, pt0, dir0, wvl, **kw):
    try:
        return rt.trace(sm, pt0, dir0, wvl, **kw), None
    except terr.TraceError as e:
        return e.ray_pkg, e


def case_rays(opm, R, rng, check_apertures, pupil_scale=1.15):
    """explicit (pt0, dir0, wvl) rays through raytrace.trace()"""
    sm, osp = opm['seq_model'], opm['optical_spec']
    wvls = list(osp['wvls'].wavelengths)
    flds = osp['fov'].fields
    N = len(sm.ifcs)
    pt0 = np.zeros((3, R))
    dir0 = np.zeros((3, R))
    wi = rng.integers(0, len(wvls), R).astype(np.int32)
    out = SoA(N, R)
    for r in range(R):
        fld = flds[rng.integers(0, len(flds))]
        pupil = rng.uniform(-pupil_scale, pupil_scale, 2)
        p, d = osp.ray_start_from_osp(pupil, fld, 'rel pupil')
        if d[2] * sm.z_dir[0] < 0:
            d = -d
        pt0[:, r], dir0[:, r] = p, d
        pkg, err = trace_with_errors(sm, p.copy(), d.copy(), wvls[wi[r]],
                                     check_apertures=check_apertures)
        out.put(r, pkg, err)
    d = dict(pt0=pt0, dir0=dir0, wvl_idx=wi,
             flags=np.int64(abi.INTERSECT_OBJ |
                            (abi.CHECK_APERTURES if check_apertures else 0)),
             first_surf=np.int64(1), last_surf=np.int64(N - 2))
    d.update(out.arrays())
    return d


class Recorder:
    """wraps rayoptics.raytr.trace.trace_safe to keep every RayResult and the
    (vignetted, mutated-in-place) pupil array the driver passed."""

    def __init__(self):
        self.results = []
        self._orig = trace.trace_safe

    def __enter__(self):
        def wrapped(opt_model, pupil, *a, **k):
            res = self._orig(opt_model, pupil, *a, **k)
            self.results.append((np.array(pupil, dtype=float), res))
            return res
        trace.trace_safe = wrapped
        return self

    def __exit__(self, *exc):
        trace.trace_safe = self._orig


def case_grid(opm, fi, wvl, num, kind='grid', start=(-1., -1.), stop=(1., 1.)):
    """the reference's trace_grid / trace_fan driver loop, FULL packets"""
    sm, osp = opm['seq_model'], opm['optical_spec']
    fld = osp['fov'].fields[fi]
    foc = osp['focus'].focus_shift
    N = len(sm.ifcs)
    rng_def = [np.array(start), np.array(stop), num]
    with Recorder() as rec:
        if kind == 'grid':
            # img_filter returns a scalar so that np.array(grid) at
            # trace.py:605 stays homogeneous under NumPy 2 (ragged partial
            # packets otherwise raise); the packets are kept by the Recorder
            trace.trace_grid(opm, rng_def, fld, wvl, foc,
                             img_filter=lambda p, pkg: 0.0,
                             form='list', append_if_none=True,
                             rayerr_filter='full')
        else:
            trace.trace_fan(opm, rng_def, fld, wvl, foc, img_filter=None,
                            rayerr_filter='full', check_apertures=True)
    R = len(rec.results)
    out = SoA(N, R)
    pupil = np.zeros((2, R))
    for r, (pup, res) in enumerate(rec.results):
        pupil[:, r] = pup
        out.put(r, res.pkg, res.err)
    d = dict(field=field_arr(field_from_model(opm, fld)),
             wvl_idx=np.int64(list(osp['wvls'].wavelengths).index(wvl)),
             start=np.array(start), stop=np.array(stop), num=np.int64(num),
             kind=np.int64(abi.GRID_PRODUCT if kind == 'grid' else abi.GRID_FAN),
             flags=np.int64(abi.INTERSECT_OBJ | abi.CHECK_APERTURES |
                            abi.APPLY_VIGNETTING),
             first_surf=np.int64(1), last_surf=np.int64(N - 2), pupil=pupil)
    d.update(out.arrays())
    return d


def case_spot(opm, num_rays):
    """SpotDiagramFigure's own data path: per field, per wavelength, the
    (R_ok, 2) arrays of transverse aberrations"""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    from rayoptics.mpl.axisarrayfigure import SpotDiagramFigure
    osp = opm['optical_spec']
    fig = plt.figure(FigureClass=SpotDiagramFigure, opt_model=opm,
                     num_rays=num_rays)
    fig.update_data()
    d = dict(num=np.int64(num_rays), foc=np.float64(osp['focus'].focus_shift))
    nf = len(osp['fov'].fields)
    # axis_data_array is filled in reversed(range(num_rows)) order
    for row_i, row in enumerate(fig.axis_data_array):
        fi = nf - 1 - row_i
        grids, _max_val, _rc = row[0]
        fld = osp['fov'].fields[fi]
        d[f'f{fi}_field'] = field_arr(field_from_model(opm, fld))
        d[f'f{fi}_image_pt'] = np.array(fld.ref_sphere[0][:2])
        for wi, g in enumerate(grids):
            d[f'f{fi}_w{wi}_hits'] = np.array(g, dtype=float).reshape(-1, 2)
    plt.close(fig)
    return d


def case_list_of_rays(opm, R, rng):
    """analyses.trace_list_of_rays with output_filter='last'"""
    sm, osp = opm['seq_model'], opm['optical_spec']
    wvls = list(osp['wvls'].wavelengths)
    fld = osp['fov'].fields[-1]
    rays = []
    for r in range(R):
        pupil = rng.uniform(-0.9, 0.9, 2)
        p, d = osp.ray_start_from_osp(pupil, fld, 'rel pupil')
        rays.append((p, d, wvls[r % len(wvls)]))
    res = analyses.trace_list_of_rays(opm, rays, output_filter='last',
                                      rayerr_filter='summary',
                                      check_apertures=True)
    last = np.full((10, R), np.nan)
    op = np.full(R, np.nan)
    status = np.zeros(R, dtype=np.uint8)
    for r, item in enumerate(res):
        if isinstance(item[1], terr.TraceError):
            status[r] = status_of(item[1])
        else:
            seg, op_r, _w = item
            last[0:3, r], last[3:6, r], last[6, r], last[7:10, r] = seg[0], seg[1], seg[2], seg[3]
            op[r] = op_r
    return dict(pt0=np.array([r[0] for r in rays]).T.copy(),
                dir0=np.array([r[1] for r in rays]).T.copy(),
                wvl_idx=np.array([i % len(wvls) for i in range(R)], dtype=np.int32),
                last=last, op=op, status=status)


def case_opd(opm, fi, wvl, num):
    """analyses.eval_wavefront (analyses.py:699-732): OPD in waves over the
    vignetted pupil bounding box, every ray through wave_abr_full_calc"""
    from rayoptics_amd.table import wavefront_from_model, wavefront_to_array
    sm, osp = opm['seq_model'], opm['optical_spec']
    fld = osp['fov'].fields[fi]
    foc = osp['focus'].focus_shift
    N = len(sm.ifcs)
    grid = analyses.eval_wavefront(opm, fld, wvl, foc, num_rays=num)
    vig_bbox = fld.vignetting_bbox(osp['pupil'], oversize=1.)
    return dict(field=field_arr(field_from_model(opm, fld)),
                wvl_idx=np.int64(list(osp['wvls'].wavelengths).index(wvl)),
                start=np.array(vig_bbox[0]), stop=np.array(vig_bbox[1]), num=np.int64(num),
                flags=np.int64(abi.INTERSECT_OBJ | abi.CHECK_APERTURES),
                first_surf=np.int64(1), last_surf=np.int64(N - 2),
                wavefront=wavefront_to_array(wavefront_from_model(opm, fld)),
                convert_to_opd=np.float64(1 / opm.nm_to_sys_units(wvl)),
                opd_grid=np.array(grid, dtype=float))


def save(name, table, cases):
    flat = {'table_json': np.array(json.dumps(table.to_dict()))}
    for cname, d in cases.items():
        for k, v in d.items():
            flat[f'{cname}/{k}'] = v
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **flat)
    print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB, cases {list(cases)}')


def kat_dblgauss_seq():
    """the reference's only end-to-end KAT: test_sequential.py + marginal_ray.py
    (CODE V axial marginal ray, 6 decimals) on gen_sequence(ag_dblgauss)."""
    import copy
    sys.path.insert(0, os.path.join(rm.REF_SRC, 'rayoptics', 'raytr', 'tests'))
    import ag_dblgauss_s as dblg
    import marginal_ray as f1r2
    from rayoptics.util.misc_math import normalize
    d = copy.deepcopy(dblg.ag_dblgauss)
    d[-2][1] += d[-1][1]            # test_sequential.py:26-27
    d[-1][1] = 0.
    # the CODE V truth was computed with the listed n_d at 587.6 nm: keep the
    # indices exactly as listed (constant-index media), not through the shim's
    # synthetic dispersion
    d = [row[:3] for row in d]
    path = list(gen_sequence(d, wvl=587.6, radius_mode=False))
    table = ra.SurfaceTable.from_paths([path], [587.6])
    p0 = np.array([0., 0., 0.])
    p1 = np.array([0., 0., d[0][1]])
    epd = np.array([25., 0., 0.])
    d0 = normalize((p1 + epd) - p0)
    ray, op, _ = rt.trace_raw(iter(path), p0, d0, 587.6)
    out = SoA(len(path), 1)
    out.put(0, (ray, op, 587.6), None)
    codev = np.array([[*r[0], *r[1], r[2]] for r in f1r2.rayf1r2])
    # a random bundle through the same hand-built path (no apertures)
    rng = np.random.default_rng(SEED)
    R = 256
    tgt = np.stack([rng.uniform(-32, 32, R), rng.uniform(-32, 32, R),
                    np.full(R, d[0][1])])
    pt0 = np.zeros((3, R))
    dir0 = tgt / np.linalg.norm(tgt, axis=0)
    bundle = SoA(len(path), R)
    for r in range(R):
        try:
            pkg, err = rt.trace_raw(iter(path), pt0[:, r].copy(), dir0[:, r].copy(),
                                    587.6, first_surf=1, last_surf=11), None
        except terr.TraceError as e:
            pkg, err = e.ray_pkg, e
        bundle.put(r, pkg, err)
    cases = {'marginal': dict(pt0=p0.reshape(3, 1), dir0=d0.reshape(3, 1),
                              codev=codev, **out.arrays()),
             'bundle': dict(pt0=pt0, dir0=dir0, first_surf=np.int64(1),
                            last_surf=np.int64(11), **bundle.arrays())}
    save('dblgauss_seq', table, cases)


def workload_file(opm, name, desc):
    """ray-optics_amd/data/<name>.json: what bench.py / smoke() need to run a
    BASELINE.json configuration where the reference is absent (the GPU box):
    the surface table plus, per field, the ray-start constants and the
    central-wavelength chief-ray image point (fld.ref_sphere[0], set by
    SequentialModel.trace_grid -> trace.setup_pupil_coords)."""
    sm, osp = opm['seq_model'], opm['optical_spec']
    table = ra.SurfaceTable.from_seq_model(sm)
    foc = osp['focus'].focus_shift
    wvl = sm.central_wavelength()
    flds, aim = [], []
    fod = opm['analysis_results']['parax_data'].fod
    for fld in osp['fov'].fields:
        rs_pkg, cr_pkg = trace.setup_pupil_coords(opm, fld, wvl, foc)
        fld.chief_ray, fld.ref_sphere = cr_pkg, rs_pkg
        fs = field_from_model(opm, fld)
        flds.append(dict(field=field_arr(fs).tolist(), kind=int(fs.kind),
                         cr_dir=[float(v) for v in fs.cr_dir], rot=[float(v) for v in fs.rot],
                         rot_order=int(fs.rot_order),
                         image_pt=[float(v) for v in rs_pkg[0][:2]]))
        # the chief-ray aiming problem the reference solved for this field
        # (trace.aim_chief_ray -> iterate_ray, trace.py:313-415, 627-640) and the
        # aim point it converged to, recomputed here from a clean start
        pt0, _d0 = osp.obj_coords(fld)
        aim_ref = trace.aim_chief_ray(opm, fld, wvl)
        if pt0[0] == 0.0 and sm.stop_surface is not None and not osp['fov'].is_wide_angle:
            aim.append(dict(pt0=[float(v) for v in pt0],
                            z_enp=float(fod.obj_dist + fod.enp_dist),
                            z_dir0=float(sm.z_dir[0]), wvl_idx=table.wvl_index(wvl),
                            surf=int(sm.stop_surface), aim_y=float(aim_ref[1])))
    # the 2-D branch of iterate_ray (fields off the y axis: scipy.optimize.fsolve = MINPACK
    # hybrd, trace.py:394-410): the same fields moved off axis in x, solved by the reference
    aim2d = []
    if sm.stop_surface is not None and not osp['fov'].is_wide_angle:
        rng2 = np.random.default_rng(2025)
        for fld in osp['fov'].fields:
            keep_xy = (fld.x, fld.y)
            for _trial in range(3):
                fld.x = float(rng2.uniform(-0.8, 0.8))
                fld.y = float(rng2.uniform(-0.8, 0.8))
                pt0, _d0 = osp.obj_coords(fld)
                aim_ref = trace.aim_chief_ray(opm, fld, wvl)
                aim2d.append(dict(pt0=[float(v) for v in pt0],
                                  z_enp=float(fod.obj_dist + fod.enp_dist),
                                  z_dir0=float(sm.z_dir[0]), wvl_idx=table.wvl_index(wvl),
                                  surf=int(sm.stop_surface), epsfcn=float(0.0001 * fod.enp_radius),
                                  aim=[float(aim_ref[0]), float(aim_ref[1])]))
            fld.x, fld.y = keep_xy
    # the vignetting search the reference runs per field (vigcalc.calc_vignetting_for_field,
    # vigcalc.py:233-340) and its answers; the field's own factors are restored afterwards
    import rayoptics.raytr.vigcalc as vigcalc
    vig = []
    for fi, fld in enumerate(osp['fov'].fields):
        keep = (fld.vux, fld.vlx, fld.vuy, fld.vly)
        vigcalc.calc_vignetting_for_field(opm, fld, wvl)
        vig.append(dict(field_index=fi, wvl_idx=table.wvl_index(wvl),
                        stop=-1 if sm.stop_surface is None else int(sm.stop_surface),
                        starts=[[float(v) for v in p] for p in osp['pupil'].pupil_rays[1:5]],
                        vig=[float(fld.vux), float(fld.vlx), float(fld.vuy), float(fld.vly)]))
        fld.vux, fld.vlx, fld.vuy, fld.vly = keep
    d = dict(description=desc, table=table.to_dict(), fields=flds, foc=float(foc),
             ref_wvl_idx=int(osp['wvls'].reference_wvl), aim=aim, aim2d=aim2d, vig=vig)
    path = os.path.join(HERE, '..', '..', 'ray-optics_amd', 'data', name + '.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(d, f)
    print(f'data/{name}.json: {os.path.getsize(path) / 1024:.0f} KiB')


def ingest_tables():
    """tests/golden/ingest_tables.json: the surface tables rayoptics_amd.ingest parses
    from the reference's prescription files (which do not travel to the GPU box), so
    that the GPU tests can trace BASELINE's '.zmx import' / 'CODE V .seq' systems"""
    from rayoptics_amd import ingest
    files = {'zmx_354710': 'zemax/tests/354710-C-Zemax(ZMX).zmx',
             'zmx_acl3026u': 'elem/tests/ACL3026U-Zemax(ZMX).zmx',
             'zmx_us05831776': 'zemax/tests/US05831776-1.zmx',
             'seq_ag_dblgauss': 'codev/tests/ag_dblgauss.seq',
             'seq_rc_f16': 'codev/tests/rc_f16.seq',
             'seq_threemir': 'codev/tests/threemir.seq',            # decentered + tilted mirrors
             'seq_codv_35571': 'codev/tests/CODV_35571.seq',        # off-axis parabola
             'zmx_zmax_37992': 'zemax/tests/zmax_37992.zmx',        # COORDBRK
             'zmx_hoo_ex46': 'zemax/tests/HoO-V2C18Ex46.zmx',       # COORDBRKs + TOROIDAL
             'roa_ritchey_chretien': 'models/Ritchey_Chretien.roa',
             'roa_cell_phone': 'optical/tests/cell_phone_camera.roa'}
    out = {}
    for key, rel in files.items():
        pres = ingest.read(os.path.join(rm.REF_SRC, 'rayoptics', rel))
        tbl = pres.to_table(index_of=ingest.nominal_index)
        out[key] = dict(source=rel, table=tbl.to_dict())
    path = os.path.join(HERE, 'ingest_tables.json')
    with open(path, 'w') as f:
        json.dump(out, f)
    print(f'ingest_tables.json: {os.path.getsize(path) / 1024:.0f} KiB, {list(out)}')


def psf_cases():
    """analyses.calc_psf (analyses.py:848-875) run by the reference on OPD grids its own
    eval_wavefront produced, and on synthetic grids with exact zeros / NaN / a maxdim that
    is not a power of two"""
    out = {}
    opm = rm.dblgauss()
    osp = opm['osp']
    for tag, fi, wvl, ndim, maxdim in (('dblgauss_f0', 0, 587.6, 32, 128),
                                       ('dblgauss_f2', 2, 486.1, 64, 256),
                                       ('dblgauss_f1_odd_size', 1, 656.3, 24, 90)):
        fld = osp['fov'].fields[fi]
        # RayGrid.update_data: grid = np.rollaxis(eval_wavefront(...), 2); grid[2] is the OPD plane
        opd = np.array(np.rollaxis(analyses.eval_wavefront(opm, fld, wvl, 0.0, num_rays=ndim), 2)[2],
                       dtype=float)
        out[f'{tag}/opd'] = opd
        out[f'{tag}/dims'] = np.array([ndim, maxdim])
        out[f'{tag}/psf'] = analyses.calc_psf(opd, ndim, maxdim)
        # what the device needs to reproduce that OPD grid itself (trace -> OPD -> PSF)
        from rayoptics_amd.table import wavefront_from_model, wavefront_to_array
        vig_bbox = fld.vignetting_bbox(osp['pupil'], oversize=1.)
        out[f'{tag}/field'] = field_arr(field_from_model(opm, fld))
        out[f'{tag}/wavefront'] = wavefront_to_array(wavefront_from_model(opm, fld))
        out[f'{tag}/bbox'] = np.array([vig_bbox[0], vig_bbox[1]], dtype=float)
        out[f'{tag}/wvl_idx'] = np.int64(list(osp['wvls'].wavelengths).index(wvl))
        out[f'{tag}/convert_to_opd'] = np.float64(1 / opm.nm_to_sys_units(wvl))
    out['dblgauss_table_json'] = np.array(json.dumps(
        ra.SurfaceTable.from_seq_model(opm['seq_model']).to_dict()))
    rng = np.random.default_rng(SEED + 5)
    for tag, ndim, maxdim in (('synthetic_small', 8, 12), ('synthetic_coma', 16, 50)):
        y, x = np.mgrid[-1:1:ndim * 1j, -1:1:ndim * 1j]
        r2 = x * x + y * y
        opd = 0.8 * r2 + 0.35 * (3 * r2 - 2) * y + 0.02 * rng.standard_normal((ndim, ndim))
        opd[r2 > 1.0] = np.nan
        opd[ndim // 2, ndim // 2] = 0.0             # an exact zero inside the pupil (-> phase 0)
        opd[ndim // 2 - 1, 2] = 3.0                 # an integer number of waves (phase != 1 exactly)
        out[f'{tag}/opd'] = opd
        out[f'{tag}/dims'] = np.array([ndim, maxdim])
        out[f'{tag}/psf'] = analyses.calc_psf(opd, ndim, maxdim)
    path = os.path.join(HERE, 'psf.npz')
    np.savez_compressed(path, **out)
    print(f'psf.npz: {os.path.getsize(path) / 1024:.0f} KiB, {sorted(set(k.split("/")[0] for k in out))}')


def wideangle_cases():
    """tests/golden/wideangle.npz: wide-angle pupil searches (wideangle.find_real_enp,
    wideangle.py:86-427) of two models over random field angles -- the rox_enp problem of
    every case (as the product's trace._enp_problem builds it) and what the reference
    returned: z_enp, or the fact that it raised."""
    import ctypes
    import logging
    import warnings
    import rayoptics.raytr.wideangle as wa
    from rayoptics_amd import trace as T, abi
    logging.disable(logging.CRITICAL)
    rng = np.random.default_rng(SEED + 9)
    nik = os.path.join(rm.REF_SRC, 'rayoptics', 'optical', 'tests', 'Nikon Nikkor Z 14-30mm f-4 S.roa')
    models = [('dblgauss', rm.dblgauss(), np.concatenate([[0., 14.], rng.uniform(0., 45., 38), rng.uniform(45., 89., 10)])),
              ('nikkor', rm.load_roa(nik, fov=(('object', 'angle'), 57.7), flds=[0., 30., 57.7],
                                     is_relative=False),
               np.concatenate([[0., 57.7], rng.uniform(0., 62., 38), rng.uniform(62., 89., 10)]))]
    out = {}
    for name, opm, angles in models:
        sm, osp = opm['seq_model'], opm['osp']
        fov = osp['fov']
        fov.is_wide_angle = True
        tbl = ra.SurfaceTable.from_seq_model(sm)
        stop = sm.stop_surface
        probs, z_ref, raised = [], [], []
        for k, ang in enumerate(angles):
            fld = fov.fields[-1]
            fld.x, fld.y = 0., float(ang) / (fov.value if fov.is_relative else 1.0)
            wvl = osp['wvls'].wavelengths[k % len(osp['wvls'].wavelengths)]
            # every fifth case starts from a (slightly wrong) previous answer, :128-134
            fld.aim_info = None if (k % 5 or not z_ref or raised[-1]) else z_ref[-1] * (1 + 1e-9)
            pb = T._enp_problem(opm, fld, wvl, tbl, stop)
            probs.append(np.frombuffer(ctypes.string_at(ctypes.addressof(pb), ctypes.sizeof(pb)),
                                       dtype=np.uint8).copy())
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                try:
                    z, _rr = wa.find_real_enp(opm, stop, fld, wvl)
                    z_ref.append(float(z))
                    raised.append(False)
                except Exception:
                    z_ref.append(np.nan)
                    raised.append(True)
        out[f'{name}/table_json'] = np.array(json.dumps(tbl.to_dict()))
        out[f'{name}/probs'] = np.stack(probs)
        out[f'{name}/z_enp'] = np.array(z_ref)
        out[f'{name}/raised'] = np.array(raised)
        out[f'{name}/angles'] = np.asarray(angles, dtype=float)
        print(name, len(angles), 'cases,', int(np.sum(raised)), 'where the reference raises')
    logging.disable(logging.NOTSET)
    path = os.path.join(HERE, 'wideangle.npz')
    np.savez_compressed(path, **out)
    print(f'wideangle.npz: {os.path.getsize(path) / 1024:.0f} KiB')


def _jt(obj_cls_module, obj_cls_name, attributes):
    """a class instance as json_tricks writes it (json_tricks/encoders.py class_instance_encode:
    {"__instance_type__": [module, name], "attributes": obj.__dict__})"""
    return {'__instance_type__': [obj_cls_module, obj_cls_name], 'attributes': attributes}


def _jt_array(a):
    """a NumPy array as json_tricks writes it in its default (non-compressed) mode"""
    a = np.asarray(a)
    return {'__ndarray__': a.tolist(), 'dtype': str(a.dtype), 'shape': list(a.shape), 'Corder': True}


def dump_roa(opm, path):
    """Writes the part of a .roa that the trace path reads -- seq_model (interfaces with
    inline profiles, DecenterData, clear apertures; gaps; stop; z_dir) and the optical spec --
    in json_tricks' layout, from a LIVE reference model.  json_tricks itself is absent here
    (oracle/refshim.py), so the layout is restated from its encoder: every object is its
    class path + __dict__, arrays are {"__ndarray__": ...}.  Used for the one thing no .roa
    of the reference tree carries: a non-null `decenter` record."""
    sm, osp = opm['seq_model'], opm['optical_spec']

    def profile(pf):
        at = {'cv': float(pf.cv)}
        for k in ('cc', 'ec'):
            if hasattr(pf, k):
                at[k] = float(getattr(pf, k))
        if hasattr(pf, 'coefs'):
            at['coefs'] = [float(c) for c in pf.coefs]
        return _jt(type(pf).__module__, type(pf).__name__, at)

    def decenter(d):
        if d is None:
            return None
        return _jt('rayoptics.elem.surface', 'DecenterData', {
            '_dtype': d.dtype, 'dec': _jt_array(d.dec), 'euler': _jt_array(d.euler),
            'rot_pt': _jt_array(d.rot_pt),
            'rot_mat': None if d.rot_mat is None else _jt_array(d.rot_mat)})

    def aperture(ca):
        at = {'x_offset': float(ca.x_offset), 'y_offset': float(ca.y_offset), 'rotation': 0.0,
              'is_obscuration': bool(ca.is_obscuration)}
        if type(ca).__name__ == 'Circular':
            at['radius'] = float(ca.radius)
        else:
            at['x_half_width'], at['y_half_width'] = float(ca.x_half_width), float(ca.y_half_width)
        return _jt('rayoptics.elem.surface', type(ca).__name__, at)

    def medium(m):
        name = type(m).__name__
        if name == 'Air':
            return _jt('opticalglass.opticalmedium', 'Air', {})
        if name == 'ModelGlass':
            return _jt('opticalglass.modelglass', 'ModelGlass',
                       {'n': float(m.n), 'v': float(m.vd), 'label': m.name() if callable(getattr(m, 'name', None)) else str(m.label)})
        return _jt('opticalglass.opticalmedium', 'ConstantIndex',
                   {'n': float(m.rindex(550.0)), 'label': ''})
    ifcs = []
    for ifc in sm.ifcs:
        ifcs.append(_jt('rayoptics.elem.surface', 'Surface', {
            'interact_mode': ifc.interact_mode, 'delta_n': float(getattr(ifc, 'delta_n', 0.0)),
            'decenter': decenter(ifc.decenter), 'max_aperture': float(ifc.max_aperture),
            'label': getattr(ifc, 'label', ''),
            'clear_apertures': [aperture(ca) for ca in ifc.clear_apertures], 'edge_apertures': [],
            'profile': profile(ifc.profile)}))
    gaps = [_jt('rayoptics.seq.gap', 'Gap', {'thi': float(g.thi), 'medium': medium(g.medium)})
            for g in sm.gaps]
    wv = osp['wvls']
    fov, pup = osp['fov'], osp['pupil']
    spec = _jt('rayoptics.raytr.opticalspec', 'OpticalSpecs', {
        'spectral_region': _jt('rayoptics.raytr.opticalspec', 'WvlSpec', {
            'wavelengths': [float(w) for w in wv.wavelengths],
            'spectral_wts': [float(w) for w in wv.spectral_wts], 'reference_wvl': int(wv.reference_wvl)}),
        'pupil': _jt('rayoptics.raytr.opticalspec', 'PupilSpec',
                     {'key': ['aperture'] + list(pup.key), 'value': float(pup.value)}),
        'field_of_view': _jt('rayoptics.raytr.opticalspec', 'FieldSpec', {
            'key': ['field'] + list(fov.key), 'value': float(fov.value),
            'is_relative': bool(fov.is_relative),
            'fields': [_jt('rayoptics.raytr.opticalspec', 'Field', {'x': float(f.x), 'y': float(f.y)})
                       for f in fov.fields]})})
    doc = {'optical_model': _jt('rayoptics.optical.opticalmodel', 'OpticalModel', {
        'ro_version': 'written by tests/golden/make_golden.py dump_roa',
        'seq_model': _jt('rayoptics.seq.sequential', 'SequentialModel', {
            'ifcs': ifcs, 'gaps': gaps, 'stop_surface': sm.stop_surface,
            'cur_surface': sm.cur_surface, 'z_dir': [int(z) for z in sm.z_dir],
            'do_apertures': False}),
        'optical_spec': spec, 'profile_dict': {}})}
    with open(path, 'w') as f:
        json.dump(doc, f, separators=(',', ':'))


def roa_decenter_fixture():
    """tests/golden/decentered.roa + decentered_roa_table.json: refmodels.tilted_singlet()
    (DecenterData 'dec and return' on a lens surface, a 'decenter' coordinate break, apertures)
    written as a .roa, and the table the reference's own transforms give for that model."""
    opm = rm.tilted_singlet()
    path = os.path.join(HERE, 'decentered.roa')
    dump_roa(opm, path)
    tbl = ra.SurfaceTable.from_seq_model(opm['seq_model'])
    with open(os.path.join(HERE, 'decentered_roa_table.json'), 'w') as f:
        json.dump(tbl.to_dict(), f)
    # the file read back through the reference's classes gives the same table
    back = rm.load_roa(path)
    t2 = ra.SurfaceTable.from_seq_model(back['seq_model'])
    for t in (tbl, t2):         # (load_roa gives the object / image surfaces the class default)
        t.rows[0].max_aperture = t.rows[-1].max_aperture = 1.0
    same = all(bytes(a) == bytes(b) for a, b in zip(tbl.rows, t2.rows))
    print(f'decentered.roa: {os.path.getsize(path) / 1024:.0f} KiB; read back through the '
          f'reference classes: rows identical = {same}')


def time_trace_workloads():
    """ray-optics_amd/data/tt_*.json: the models of the reference's own benchmark of this path
    (rayoptics/raytr/tests/time_trace.py -> trace_results.txt) that are not BASELINE
    configurations already, through the reference's importers (refmodels.time_trace_model)"""
    import logging
    logging.disable(logging.CRITICAL)
    for name, rel, row, pub in rm.TIME_TRACE_MODELS:
        workload_file(rm.time_trace_model(rel), name,
                      f"reference benchmark model '{row}' (rayoptics/{rel}; "
                      f'rayoptics/raytr/tests/trace_results.txt: {pub} rays/s on the author\'s machine)')
    logging.disable(logging.NOTSET)


C3_ZMX_DESC = ('BASELINE.json configs[2]: Zemax .zmx import -- rayoptics/zemax/tests/US08427765-1.ZMX, '
               '13 interfaces incl. one EVENASPH, 3 real-image-height fields x 3 wavelengths, image '
               "f/2.1 -- read by the reference's own zmxread; the five catalogue glasses carry their "
               'nominal (nd, vd)')


def main():
    rng = np.random.default_rng(SEED)
    if '--only-psf' in sys.argv:
        psf_cases()
        return
    if '--only-ingest' in sys.argv or '--workloads-only' in sys.argv:
        ingest_tables()
        if '--only-ingest' in sys.argv:
            return
    if '--only-telecentric' in sys.argv:
        opm = rm.telecentric()
        save('telecentric', ra.SurfaceTable.from_seq_model(opm['seq_model']), {
            'opd_f0': case_opd(opm, 0, 550.0, 11),
            'opd_f2': case_opd(opm, 2, 486.1, 10),
        })
        return
    if '--only-time-trace' in sys.argv:
        time_trace_workloads()
        return
    if '--only-roa-decenter' in sys.argv:
        roa_decenter_fixture()
        return
    if '--only-wideangle' in sys.argv:
        wideangle_cases()
        return
    if '--only-c3-zmx' in sys.argv:
        workload_file(rm.zmx_evenasph_c3(), 'zmx_evenasph_c3', C3_ZMX_DESC)
        return
    if '--workloads-only' not in sys.argv:
        kat_dblgauss_seq()
    workload_file(rm.dblgauss(), 'dblgauss_c2',
                  'BASELINE.json configs[1]: double Gauss, 13 interfaces (K=12 '
                  'intersections/ray), rayoptics/raytr/tests/ag_dblgauss_s.py; '
                  'EPD 50, fields 0/10/14 deg, 656.3/587.6/486.1 nm')
    workload_file(rm.rc_telescope(), 'rc_telescope_c4',
                  'BASELINE.json configs[3]: Ritchey-Chretien mirror pair + field '
                  'stop, 5 fields (rayoptics/models/Ritchey_Chretien.roa)')
    workload_file(rm.singlet(), 'singlet_c1',
                  'BASELINE.json configs[0]: singlet, 4 interfaces (the shape of '
                  'rayoptics/models/singlet_f5.roa)')
    workload_file(rm.cell_phone(), 'cell_phone',
                  '13-interface phone lens, 8 RadialPolynomial aspheres '
                  '(rayoptics/optical/tests/cell_phone_camera.roa): the asphere model of '
                  'the reference timing table (rayoptics/raytr/tests/trace_results.txt:10)')
    workload_file(rm.nikkor(), 'nikkor_c3',
                  'BASELINE.json configs[2] stand-in: 29-interface zoom with 4 '
                  'even aspheres (rayoptics/optical/tests/Nikon Nikkor Z 14-30mm f-4 S.roa)')
    workload_file(rm.zmx_evenasph_c3(), 'zmx_evenasph_c3', C3_ZMX_DESC)
    workload_file(rm.litho_c5(), 'litho_c5',
                  'BASELINE.json configs[4] stand-in: the largest prescription in the reference '
                  'tree, rayoptics/zemax/tests/US05831776-1.zmx (44 interfaces, K=43, 248 nm '
                  'lithography lens, object NA 0.15), imported by the reference\'s own zmxread; '
                  '9 fields x 5 wavelengths as configs[4] asks; fused-silica indices from the '
                  'Malitson formula (rayoptics_amd.ingest.SELLMEIER)')
    if '--workloads-only' in sys.argv:
        return

    # C2: double Gauss (13 interfaces)
    opm = rm.dblgauss()
    table = ra.SurfaceTable.from_seq_model(opm['seq_model'])
    save('dblgauss', table, {
        'rays_ap': case_rays(opm, 384, rng, True),
        'rays_noap': case_rays(opm, 128, rng, False),
        'grid_f0': case_grid(opm, 0, 587.6, 12),
        'grid_f2': case_grid(opm, 2, 486.1, 12),
        'fan_f1': case_grid(opm, 1, 656.3, 21, kind='fan', start=(0., -1.), stop=(0., 1.)),
        'spot': case_spot(opm, 16),
        'list_last': case_list_of_rays(opm, 64, rng),
        'opd_f0': case_opd(opm, 0, 587.6, 15),
        'opd_f2': case_opd(opm, 2, 486.1, 16),
    })
    # finite-conjugate variant separates kernel bugs from the 1e10 cancellation
    opm = rm.dblgauss(obj_thi=1.0e3)
    save('dblgauss_finite', ra.SurfaceTable.from_seq_model(opm['seq_model']), {
        'rays_ap': case_rays(opm, 256, rng, True),
        'grid_f2': case_grid(opm, 2, 587.6, 10),
    })

    # C1: singlet, 64x64 grid through the reference trace_grid (CPU plumbing)
    opm = rm.singlet()
    save('singlet', ra.SurfaceTable.from_seq_model(opm['seq_model']), {
        'grid64': case_grid(opm, 0, 650.0, 64),
        'grid_f1': case_grid(opm, 1, 650.0, 9),
        'spot': case_spot(opm, 12),
    })

    # C4: Ritchey-Chretien mirrors + field stop
    opm = rm.rc_telescope()
    save('rc_telescope', ra.SurfaceTable.from_seq_model(opm['seq_model']), {
        'rays_ap': case_rays(opm, 256, rng, True, pupil_scale=1.3),
        'grid_f0': case_grid(opm, 0, 550.0, 10),
        'grid_f4': case_grid(opm, 4, 550.0, 10),
        'spot': case_spot(opm, 10),
        'opd_f3': case_opd(opm, 3, 550.0, 12),
    })

    # C3 stand-in: 29 interfaces, 4 even aspheres
    opm = rm.nikkor()
    save('nikkor', ra.SurfaceTable.from_seq_model(opm['seq_model']), {
        'rays_ap': case_rays(opm, 192, rng, True),
        'grid_f1': case_grid(opm, 1, 587.5618, 8),
        'spot': case_spot(opm, 8),
        'opd_f1': case_opd(opm, 1, 587.5618, 12),
    })

    # RadialPolynomial aspheres (the reference's timed asphere model)
    opm = rm.cell_phone()
    save('cell_phone', ra.SurfaceTable.from_seq_model(opm['seq_model']), {
        'rays_ap': case_rays(opm, 192, rng, True),
        'grid_f2': case_grid(opm, 2, list(opm['osp']['wvls'].wavelengths)[0], 8),
    })

    # decenters / tilts / phantom / rectangular + obscuration
    opm = rm.tilted_singlet()
    save('tilted_singlet', ra.SurfaceTable.from_seq_model(opm['seq_model']), {
        'rays_ap': case_rays(opm, 256, rng, True, pupil_scale=1.6),
        'grid_f1': case_grid(opm, 1, 650.0, 10),
        'opd_f1': case_opd(opm, 1, 550.0, 11),
    })

    # infinite reference sphere (image-space telecentric): wave_abr_full_calc_inf_ref
    opm = rm.telecentric()
    save('telecentric', ra.SurfaceTable.from_seq_model(opm['seq_model']), {
        'opd_f0': case_opd(opm, 0, 550.0, 11),
        'opd_f2': case_opd(opm, 2, 486.1, 10),
    })

    psf_cases()
    wideangle_cases()
    roa_decenter_fixture()
    time_trace_workloads()

    # aspheric toroids (Newton path, anamorphic)
    opm = rm.toroid_lens()
    save('toroid_lens', ra.SurfaceTable.from_seq_model(opm['seq_model']), {
        'rays_ap': case_rays(opm, 256, rng, True, pupil_scale=1.5),
        'grid_f1': case_grid(opm, 1, 450.0, 11),
        'opd_f1': case_opd(opm, 1, 550.0, 9),
    })


if __name__ == '__main__':
    main()

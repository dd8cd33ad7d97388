"""Drop-ins for the vignetting search of ``rayoptics.raytr.vigcalc`` and the
boundary-ray trace behind ``set_clear_apertures`` (SURVEY.md section 8f row 2:
they run on every model update and dominate interactive latency once grids are
fast).

  calc_vignetting_for_field <- rayoptics/raytr/vigcalc.py:233-256 (+ calc_vignetted_ray
                               :259-340, iterate_pupil_ray :396-461)
  set_vig                   <- rayoptics/raytr/vigcalc.py:99-105: every field's four
                               pupil directions in one launch
  iterate_pupil_ray         <- rayoptics/raytr/vigcalc.py:396-461 on its own: what
                               vigcalc.set_pupil (:123-230) finds the edge of the stop with.
                               set_pupil / set_stop_aperture (:108-121) themselves stay the
                               reference's code: everything they iterate or trace in bulk --
                               iterate_pupil_ray, set_vig, set_clear_apertures' boundary rays --
                               resolves to these drop-ins through vigcalc's module globals
  trace_boundary_rays_at_field <- rayoptics/raytr/trace.py:436-452: the pupil-limiting
                               rays of a field in one launch (set_clear_apertures,
                               vigcalc.py:45-85, consumes them unchanged)
"""
import numpy as np

from . import abi, session
from .table import field_from_model


def _problems(opt_model, fld, wvl, tbl, max_iter):
    osp = opt_model['optical_spec']
    stop = opt_model['seq_model'].stop_surface
    f = field_from_model(opt_model, fld)
    wi = tbl.wvl_index(wvl)
    probs = []
    for i, start in enumerate(osp['pupil'].pupil_rays[1:5]):
        p = abi.Vig()
        p.fld = f
        start = np.array(start, dtype=float)
        unit = start / np.linalg.norm(start) if np.linalg.norm(start) != 0.0 else start   # misc_math.normalize
        p.start_dir[0], p.start_dir[1] = start
        p.unit_dir[0], p.unit_dir[1] = unit
        p.xy = i // 2
        p.wvl_idx = wi
        p.stop_surf = -1 if stop is None else int(stop)
        p.max_iter = int(max_iter)
        probs.append(p)
    return probs


def iterate_pupil_ray(opt_model, indx, xy, start_r0, r_target, fld, wvl, **kwargs):
    """rayoptics/raytr/vigcalc.py:396-461: the pupil coordinate on axis ``xy`` whose ray meets
    interface ``indx`` at radius ``r_target`` -- scipy's secant iteration around single Python
    ray traces in the reference, one lane of one launch here (``rox_iterate_pupil_rays``).
    ``vigcalc.set_pupil`` (:123-230) iterates the axial marginal ray to the edge of the stop
    with it; the vignetting search has the same iteration inside its own launch."""
    start_coords = np.array([0., 0.])
    if indx is None:            # floating stop surface - use entrance pupil for aiming (:463-464)
        start_coords[xy] = r_target
        return start_coords
    eng = session.engine_for(opt_model)
    p = abi.PupilIter()
    p.fld = field_from_model(opt_model, fld)
    p.start_r0, p.r_target = float(start_r0), float(r_target)
    p.xy, p.wvl_idx, p.indx = int(xy), eng.table.wvl_index(wvl), int(indx)
    start_coords[xy] = eng.iterate_pupil_rays([p])[0]
    return start_coords


def _store(fld, vig4):
    fld.vux, fld.vlx, fld.vuy, fld.vly = (np.float64(v) for v in vig4)  # vigcalc.py:252-256 (np.float64 there too)


def calc_vignetting_for_field(opm, fld, wvl, **kwargs):
    """rayoptics/raytr/vigcalc.py:233-256"""
    eng = session.engine_for(opm)
    vig, _clip = eng.calc_vignetting(_problems(opm, fld, wvl, eng.table,
                                               kwargs.get('max_iter_count', 50)))
    _store(fld, vig)


def set_vig(opm, **kwargs):
    """rayoptics/raytr/vigcalc.py:99-105, all fields in one launch.  The search of
    one field never reads another field's vignetting, so batching changes nothing."""
    osp = opm['osp']
    eng = session.engine_for(opm)
    probs, flds = [], []
    for fi in range(len(osp['fov'].fields)):
        fld, wvl, _foc = osp.lookup_fld_wvl_focus(fi)
        probs += _problems(opm, fld, wvl, eng.table, kwargs.get('max_iter_count', 50))
        flds.append(fld)
    vig, _clip = eng.calc_vignetting(probs)
    for k, fld in enumerate(flds):
        _store(fld, vig[4 * k:4 * k + 4])


def trace_boundary_rays_at_field(opt_model, fld, wvl, use_named_tuples=False, **kwargs):
    """rayoptics/raytr/trace.py:436-452: a list of RayPkgs for the boundary rays of
    `fld` (partial packets for rays that fail: rayerr_filter='full')"""
    from .analyses import _setup_pupil_coords
    from .trace import _trace_pupil, emit
    kwargs['rayerr_filter'] = kwargs.get('rayerr_filter', 'full')
    ref_sphere, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, 0.0)
    fld.chief_ray = cr_pkg
    fld.ref_sphere = ref_sphere
    rayerr_filter = kwargs.pop('rayerr_filter')
    output_filter = kwargs.pop('output_filter', None)
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', True)
    pupil_rays = opt_model.optical_spec.pupil.pupil_rays
    pc = np.array([[p[0], p[1]] for p in pupil_rays], dtype=float)
    pk = _trace_pupil(opt_model, fld, wvl, kwargs, output_filter, rayerr_filter,
                      pupil_list=(pc[:, 0].copy(), pc[:, 1].copy()))
    ifcs = opt_model['seq_model'].ifcs
    return [emit(pk, r, output_filter, rayerr_filter, use_named_tuples, ifcs)[0]
            for r in range(len(pupil_rays))]


def _materialised(pkg):
    """a boundary-ray package with its segments as a real list, as the reference returns it:
    ``max_aperture_at_surf`` (vigcalc.py:31-42) indexes every segment of every rim ray for every
    interface -- 370 lazy segment builds per model update otherwise"""
    if pkg is None:
        return None
    ray = pkg[0]
    if hasattr(ray, 'to_list'):
        ray = ray.to_list()
        return pkg._replace(ray=ray) if hasattr(pkg, '_replace') else (ray,) + tuple(pkg[1:])
    return pkg


_RIM = ((0., 0.), (1., 0.), (-1., 0.), (0., 1.), (0., -1.))    # PupilSpec.default_pupil_rays
_RIM_IN_3X3 = (4, 7, 1, 5, 3)                                     # ray r = i * 3 + j of x = -1 + i, y = -1 + j


def trace_boundary_rays(opt_model, **kwargs):
    """rayoptics/raytr/trace.py:467-475: the boundary rays of EVERY field at the central
    wavelength (``set_clear_apertures`` asks for them on every model update,
    vigcalc.py:68).  The reference traces five single rays per field; the per-field drop-in
    (:func:`trace_boundary_rays_at_field`) made that one launch per field; here all fields are
    ONE launch: the five standard pupil points are points of the 3 x 3 product grid over
    [-1, 1]^2 (whose accumulated coordinates -1, -1 + 1, -1 + 1 + 1 are exactly -1, 0, 1), so
    the batch is ``trace_pupil_grids_host`` of that grid for every field -- FULL packets
    straight into pinned host memory -- and rays 4, 7, 1, 5, 3 of each item are the rim rays.
    Anything else (other pupil rays, extra keywords) takes the per-field route."""
    import rayoptics.raytr.trace as rtrace
    from . import abi
    from .analyses import _setup_pupil_coords
    from .engine import make_grid
    from .raypkg import HostPackets
    from .table import field_from_model
    from .trace import emit, opts_from_kwargs
    osp = opt_model.optical_spec
    fov = osp.field_of_view
    wvl = opt_model.seq_model.central_wavelength()
    pupil_rays = osp.pupil.pupil_rays
    std = (len(pupil_rays) == 5 and
           all(float(p[0]) == q[0] and float(p[1]) == q[1] for p, q in zip(pupil_rays, _RIM)))
    if not std or (set(kwargs) - {'use_named_tuples', 'rayerr_filter'}):
        rayset = []
        for fld in fov.fields:
            rim_rays = rtrace.trace_boundary_rays_at_field(opt_model, fld, wvl, **kwargs)
            fld.pupil_rays = rtrace.boundary_ray_dict(opt_model, rim_rays)
            rayset.append(rim_rays)
        return rayset
    named = kwargs.get('use_named_tuples', False)
    rayerr_filter = kwargs.get('rayerr_filter', 'full')
    with session.hold(opt_model) as eng:
        for fld in fov.fields:                  # (may aim a field that has no chief ray yet)
            ref_sphere, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, 0.0)
            fld.chief_ray = cr_pkg
            fld.ref_sphere = ref_sphere
        tbl = eng.table
        wi = tbl.wvl_index(wvl)
        oc = eng.memo.obj_coords
        flds, optl = [], []
        for fld in fov.fields:
            f = field_from_model(opt_model, fld, 'rel pupil', cache=oc)
            opts = opts_from_kwargs(tbl.n_ifcs, {'apply_vignetting': True}, abi.OUT_FULL)
            if f.kind == abi.FLD_EPD_WIDE or f.z_dir0 == 0.0:
                opts.flags &= ~abi.INTERSECT_OBJ        # trace.py:302-303
            flds.append(f)
            optl.append(opts)
        res = eng.trace_pupil_grids_host(flds, [wi] * len(flds), make_grid((-1., -1.), (1., 1.), 3), optl)
        ifcs = opt_model['seq_model'].ifcs
        rayset = []
        for fld, opts, h in zip(fov.fields, optl, res):
            pk = HostPackets(h, tbl, opts.flags, abi.OUT_FULL, wvl)
            rim_rays = [_materialised(emit(pk, r, None, rayerr_filter, named, ifcs)[0]) for r in _RIM_IN_3X3]
            fld.pupil_rays = rtrace.boundary_ray_dict(opt_model, rim_rays)
            rayset.append(rim_rays)
    return rayset

"""Drop-ins for the fan / grid drivers of ``rayoptics.raytr.trace`` and for
``SequentialModel.trace_grid``: same names, arguments and return shapes, with
the per-ray Python loop replaced by one device launch.

  trace_grid      <- rayoptics/raytr/trace.py:563-605
  trace_fan       <- rayoptics/raytr/trace.py:537-560
  seq_trace_grid  <- rayoptics/seq/sequential.py:1058-1085 (method)
  raytrace_trace_raw <- rayoptics/raytr/raytrace.py:83-264 trace_raw() on an explicit path list
  raytrace_trace  <- rayoptics/raytr/raytrace.py:51-80 trace(): one ray per call, for the
                     reference's own iterative callers (trace_base, iterate_ray's 2-D
                     fsolve branch, the wide-angle pupil search, trace_chief_ray ...)
  aim_chief_ray   <- rayoptics/raytr/trace.py:627-640 (iterate_ray's 1-D branch on the device)
  trace_ray_list_at_field <- rayoptics/raytr/trace.py:478-486 (trace_field / trace_all_fields:
                     a field's boundary rays in one launch)
  trace_astigmatism <- rayoptics/raytr/trace.py:823-863 (five rays, one launch; the field loop
                     of trace_astigmatism_curve calls it)
  trace_chief_ray, trace_astigmatism_coddington_fan <- rayoptics/raytr/trace.py:513-535, 708-712
                     (the chief rays of every field x wavelength of a model state in one launch)
  iterate_ray_raw <- rayoptics/raytr/trace.py:866-961 (the reverse chief ray of
                     wideangle.eval_real_image_ht: the whole iteration in one launch)
  osp_update_optical_properties <- rayoptics/raytr/opticalspec.py:263-281 (method; all
                                   fields aimed in one launch)
Result filtering follows trace_safe, rayoptics/raytr/trace.py:160-221.
"""
import threading

import numpy as np

from . import abi, session
from .engine import make_opts, make_grid
from .raypkg import HostPackets, RayPkg, RaySeg
from .table import field_from_model, UnsupportedModelError
from .traceerror import TraceError

# what table.field_from_model raises for a field this layer cannot express (its docstring): such
# a field is answered by the reference's own function.  Anything else -- an AttributeError, a
# NameError: a programming error -- propagates.
_FIELD_ERRORS = (UnsupportedModelError, TraceError, TypeError, ValueError, ZeroDivisionError,
                 FloatingPointError)


def opts_from_kwargs(n_ifcs, kwargs, out_mode, foc=0.0, image_pt=(0., 0.), wf=None):
    """kwargs threaded through the reference's layers (trace.py:116-146,
    raytrace.py:83-99) -> the C ABI's flags word + scalars.  first/last_surf
    default as in raytrace.trace (raytrace.py:77-79)."""
    flags = 0
    if kwargs.get('check_apertures', False):
        flags |= abi.CHECK_APERTURES
    if kwargs.get('intersect_obj', True):
        flags |= abi.INTERSECT_OBJ
    if kwargs.get('filter_out_phantoms', False):
        flags |= abi.FILTER_PHANTOMS
    if kwargs.get('apply_vignetting', False):
        flags |= abi.APPLY_VIGNETTING
    if session.TOLERANCE_MODE:                  # (the library takes it for reduced-output modes only)
        flags |= abi.FAST_FP64
    last = kwargs.get('last_surf', n_ifcs - 2)
    fuzz = kwargs.get('pt_inside_fuzz', None)
    return make_opts(flags=flags, out_mode=out_mode,
                     first_surf=kwargs.get('first_surf', 1),
                     last_surf=-1 if last is None else last,
                     eps=kwargs.get('eps', 1.0e-12),
                     fuzz=1e-5 if fuzz is None else fuzz, foc=foc, image_pt=image_pt, wf=wf)


def emit(pk, r, output_filter, rayerr_filter, named, ifcs):
    """(pkg, err) for ray r, as trace_safe would return it (trace.py:186-221)"""
    if pk.status_of(r) != abi.OK:
        if rayerr_filter == 'full':
            err = pk.error(r, ifcs, with_pkg=True, named=True)
            return err.ray_pkg, err
        if rayerr_filter == 'summary':
            return None, pk.error(r, ifcs, with_pkg=False)
        return None, None
    pkg = pk.pkg(r, named)
    if output_filter is None:
        return pkg, None
    if output_filter == 'last':
        ray, op, wvl = pkg
        return RayPkg([ray[-1]], op, wvl), None
    return output_filter(pkg), None


def raytrace_trace(seq_model, pt0, dir0, wvl, **kwargs):
    """rayoptics/raytr/raytrace.py:51-80 ``trace``: one ray through the model;
    returns ``(ray, op_delta, wvl)`` with ``ray`` a list of ``[p, d, dst, nrml]``
    per interface, or raises the TraceError the reference raises (``.surf``,
    ``.ifc``, ``.int_pt``, the partial ``.ray_pkg``, raytrace.py:231-257).

    Rebinding this one function puts every per-ray caller the reference still
    drives from Python -- ``trace_base``/``trace_safe``, scipy's MINPACK iteration
    in ``iterate_ray`` (fields off the y axis), the wide-angle pupil search
    (``wideangle.py:46-83``), ``SequentialModel.trace`` -- on the device trace,
    bit-identical to the reference's own ``trace_raw``."""
    eng = session.engine_for(seq_model.opt_model)
    tbl = eng.table
    wv = seq_model.central_wavelength() if wvl is None else wvl
    wi = tbl.wvl_index(wv)
    opts = opts_from_kwargs(tbl.n_ifcs, kwargs, abi.OUT_FULL)
    opts.flags &= ~abi.APPLY_VIGNETTING             # a pupil-level notion (trace.py:291-295)
    h = eng.trace_one(pt0, dir0, wi, opts)
    if h.status[0] == abi.OK:
        # the packet block is this call's own: the segments view it, no copies
        return ([[s[0:3], s[3:6], float(s[6]), s[7:10]] for s in h.seg[:, :, 0]],
                float(h.op[0]), wvl)
    pk = HostPackets(h, tbl, opts.flags, abi.OUT_FULL, wv)
    err = pk.error(0, getattr(seq_model, 'ifcs', None), with_pkg=True, named=False)
    ray, op, _w = err.ray_pkg
    err.ray_pkg = (ray.to_list(), op, wvl)
    raise err


def raytrace_trace_raw(path, pt0, dir0, wvl, eps=1.0e-12, check_apertures=False,
                       intersect_obj=True, filter_out_phantoms=False, **kwargs):
    """rayoptics/raytr/raytrace.py:83-264 ``trace_raw``: one ray through an explicit path
    -- any iterable of ``(Intfc, Gap, Tfrm, Index, Z_Dir)`` tuples: ``gen_sequence`` output
    (the reference's own unit test, raytr/tests/test_sequential.py), a reversed path
    (``iterate_ray_raw``, wideangle.py:646).  The path is flattened to a surface table on
    every call (it may be a one-shot generator) and the device handle is cached by the
    table's bytes; first_surf / last_surf default as in trace_raw itself (0, None)."""
    from .table import SurfaceTable, UnsupportedModelError
    path = list(path)
    try:
        tbl = SurfaceTable.from_paths([path], [0.0 if wvl is None else float(wvl)])
    except UnsupportedModelError:
        # the path may have been a one-shot iterator: under install('reference') the
        # reference's own trace_raw gets the materialised list, not the spent iterator
        if session.FALLBACK == 'reference':
            from . import install
            import rayoptics.raytr.raytrace as rraytrace
            theirs = install._saved.get((rraytrace, 'trace_raw'))
            if theirs is not None:
                return theirs(iter(path), pt0, dir0, wvl, eps=eps, check_apertures=check_apertures,
                              intersect_obj=intersect_obj,
                              filter_out_phantoms=filter_out_phantoms, **kwargs)
        raise
    eng = session.engine_for_table(tbl)
    flags = (abi.CHECK_APERTURES if check_apertures else 0) | \
        (abi.INTERSECT_OBJ if intersect_obj else 0) | \
        (abi.FILTER_PHANTOMS if filter_out_phantoms else 0)
    last = kwargs.get('last_surf', None)
    fuzz = kwargs.get('pt_inside_fuzz', None)
    opts = make_opts(flags=flags, out_mode=abi.OUT_FULL, first_surf=kwargs.get('first_surf', 0),
                     last_surf=-1 if last is None else last, eps=eps,
                     fuzz=1e-5 if fuzz is None else fuzz)
    h = eng.trace_one(pt0, dir0, 0, opts)
    if h.status[0] == abi.OK:
        return ([[s[0:3], s[3:6], float(s[6]), s[7:10]] for s in h.seg[:, :, 0]],
                float(h.op[0]), wvl)
    pk = HostPackets(h, tbl, opts.flags, abi.OUT_FULL, tbl.wvls[0])
    err = pk.error(0, [seg[0] for seg in path], with_pkg=True, named=False)
    ray, op, _w = err.ray_pkg
    err.ray_pkg = (ray.to_list(), op, wvl)
    raise err


def iterate_ray_raw(pthlist, ifcx, xy_target, pt0, d0, obj2pup_dist, eprad, wvl, not_wa, **kwargs):
    """rayoptics/raytr/trace.py:866-961 ``iterate_ray_raw``: the chief-ray iteration over an
    explicit path list -- the reversed path along which ``wideangle.eval_real_image_ht``
    (wideangle.py:620-665; ``FieldSpec.obj_coords`` of fields given as real image heights,
    BASELINE configs[2]'s .zmx) sends a ray back from the image point through the centre of
    the stop.  The reference runs scipy's secant / MINPACK iteration around single Python
    ray traces; here the whole iteration is one lane of one launch (``rox_iterate_ray_raw``)
    on the table of that path, and the ``rr`` the reference hands back -- the RayResult of the
    LAST TRIAL RAY it evaluated, which eval_real_image_ht reads the object-space ray from --
    is that trial ray traced once more.  Returns ``(start_coords, rr)`` as the reference."""
    from rayoptics.raytr import RayPkg as RefRayPkg, RayResult
    from rayoptics.raytr.traceerror import TraceError
    from rayoptics.util.misc_math import normalize
    from .table import SurfaceTable
    if ifcx is None:            # floating stop surface - use entrance pupil for aiming (:957-958)
        return np.array([0., 0.]) + xy_target, None
    path = list(pthlist)
    tbl = SurfaceTable.from_paths([path], [0.0 if wvl is None else float(wvl)])
    eng = session.engine_for_table(tbl)
    z_dir0 = path[0][4]
    a = abi.Aim()
    for k in range(3):
        a.pt0[k] = float(pt0[k])
    a.z_enp = float(obj2pup_dist)
    a.x_target, a.y_target = float(xy_target[0]), float(xy_target[1])
    a.z_dir0 = float(z_dir0) if z_dir0 is not None else 1.0
    a.wvl_idx, a.surf, a.flip = 0, int(ifcx), 1 if not_wa else 0
    a.two_d = 0 if (pt0[0] == 0.0 and xy_target[0] == 0.0) else 1
    a.epsfcn = 0.0001 * float(eprad)
    aim, _res, last_xy, _last_st = eng.iterate_ray_raw([a], eps=kwargs.get('eps', 1.0e-12))
    start_coords = aim[0].copy() if a.two_d else np.array([0., aim[0, 1]])
    # `rr`: the last trial ray the iteration evaluated (y_stop_coordinate / surface_coordinate,
    # :879-921), with the direction formed as the reference forms it
    pt1 = np.array([last_xy[0, 0], last_xy[0, 1], obj2pup_dist])
    dir0 = normalize(pt1 - pt0)
    if not_wa and dir0[2] * z_dir0 < 0:
        dir0 = -dir0
    try:
        rr = RayResult(RefRayPkg(*raytrace_trace_raw(iter(path), pt0, dir0, wvl)), None)
    except TraceError as ray_error:
        rr = RayResult(RefRayPkg(*ray_error.ray_pkg), ray_error)
    return start_coords, rr


def _chief_rays(opt_model, eng):
    """Chief-ray packets of EVERY (field, wavelength) of the model's current state in one
    launch (``TraceEngine.trace_pupil_grids_host``: FULL packets of a 2 x 2 pupil grid whose
    first ray is pupil (0, 0), written straight into pinned host memory), kept on the engine
    -- a model edit makes a new engine -- under the field's ray-start constants: a re-aimed
    field simply misses.  {(field constants, wavelength index): (seg [N, 10], op) | None}"""
    memo = eng.memo                                  # engine.ModelMemo, cleared by eng.close()
    cache = memo.chief_rays
    osp = opt_model['optical_spec']
    tbl = eng.table
    oc = memo.obj_coords
    kw = {'apply_vignetting': True}                 # trace_base's default; (0, 0) is its fixed point
    flds, wis, optl, keys = [], [], [], []
    for fld in osp['fov'].fields:
        try:
            f = field_from_model(opt_model, fld, 'rel pupil', cache=oc)
        except _FIELD_ERRORS:                       # (that field goes the reference's own way)
            continue
        opts = opts_from_kwargs(tbl.n_ifcs, kw, abi.OUT_FULL)
        if f.kind == abi.FLD_EPD_WIDE or f.z_dir0 == 0.0:
            opts.flags &= ~abi.INTERSECT_OBJ        # trace.py:302-303
        fb = bytes(f)
        for wi in range(len(tbl.wvls)):
            if (fb, wi) in cache:
                continue
            flds.append(f)
            wis.append(wi)
            optl.append(opts)
            keys.append((fb, wi))
    if flds:
        res = eng.trace_pupil_grids_host(flds, wis, make_grid((0., 0.), (1., 1.), 2), optl)
        with memo.lock:
            for key, h in zip(keys, res):
                if int(h.status[0]) == abi.OK:
                    cache[key] = (np.array(h.seg[:, :, 0]), float(h.op[0]))
                else:
                    cache[key] = None               # a failing chief ray: the reference's own path
            if len(cache) > 4096:                   # (a session that re-aims for ever)
                for k in list(cache)[:len(cache) - 2048]:
                    cache.pop(k, None)
    return cache


def trace_chief_ray(opt_model, fld, wvl, foc):
    """rayoptics/raytr/trace.py:513-535 ``trace_chief_ray``: the chief ray of ``fld`` at ``wvl``
    and its exit-pupil segment.  Every analysis reaches it through ``setup_pupil_coords`` ->
    ``get_chief_ray_pkg`` (trace.py:608-624, 660-687; both stay the reference's own code, as
    does ``calculate_reference_sphere``) -- per (field, wavelength[, fan]) of a figure, each
    time one ray through the one-ray seam.  Here the first request of a model state traces
    the chief rays of ALL fields and wavelengths in one launch; the others are served from
    that batch.  The packet is the one ``trace_safe(opt_model, [0., 0.], fld, wvl,
    output_filter=None, rayerr_filter='full')`` returns; ``transfer_to_exit_pupil`` is the
    reference's own.  A wavelength outside the spectral list, a field this layer cannot
    express or a failing chief ray go to the reference's function."""
    import rayoptics.raytr.trace as rtrace
    from rayoptics.raytr import RayPkg as RefRayPkg
    pkg = _chief_ray_packet(opt_model, fld, wvl)
    if pkg is None:
        from . import install
        theirs = install._saved.get((rtrace, 'trace_chief_ray'))
        if theirs is None:
            raise RuntimeError('trace_chief_ray: not installed over the reference')
        return theirs(opt_model, fld, wvl, foc)
    seg, op = pkg
    blk = seg.copy()            # this call's own block: the segments view it (as raytrace_trace's do)
    cr = RefRayPkg([[s[0:3], s[3:6], float(s[6]), s[7:10]] for s in blk], op, wvl)
    fod = opt_model['analysis_results']['parax_data'].fod
    cr_exp_seg = rtrace.transfer_to_exit_pupil(opt_model.seq_model.ifcs[-2],
                                               (cr.ray[-2][0], cr.ray[-2][1]), fod.exp_dist)
    return cr, cr_exp_seg


def _chief_ray_packet(opt_model, fld, wvl):
    """(seg [N, 10], op) of the chief ray of (fld, wvl) from the model state's batch
    (:func:`_chief_rays`), or None where the reference's own path must answer"""
    eng = session.engine_for(opt_model)
    try:
        wi = eng.table.wvl_index(wvl)
        f = field_from_model(opt_model, fld, 'rel pupil', cache=eng.memo.obj_coords)
    except _FIELD_ERRORS + (KeyError,):             # (KeyError: a wavelength outside the spectral list)
        return None
    key = (bytes(f), wi)
    cache = eng.memo.chief_rays
    if key not in cache:
        cache = _chief_rays(opt_model, eng)
    return cache.get(key)


def trace_astigmatism_coddington_fan(opt_model, fld, wvl, foc):
    """rayoptics/raytr/trace.py:708-712: astigmatism by a Coddington trace along the chief ray
    of ``fld`` -- ``trace_ray(opt_model, [0., 0.], fld, wvl)`` (named tuples) followed by
    ``trace_coddington_fan``, which is packet algebra and stays the reference's own.  The chief
    ray comes from the model state's one-launch batch instead of the one-ray seam."""
    import rayoptics.raytr.trace as rtrace
    from rayoptics.raytr import RayPkg as RefRayPkg, RaySeg as RefRaySeg
    pkg = _chief_ray_packet(opt_model, fld, wvl)
    if pkg is None:
        from . import install
        theirs = install._saved.get((rtrace, 'trace_astigmatism_coddington_fan'))
        if theirs is None:
            raise RuntimeError('trace_astigmatism_coddington_fan: not installed over the reference')
        return theirs(opt_model, fld, wvl, foc)
    seg, op = pkg
    blk = seg.copy()
    cr = RefRayPkg([RefRaySeg(s[0:3], s[3:6], float(s[6]), s[7:10]) for s in blk], op, wvl)
    return rtrace.trace_coddington_fan(opt_model, cr, foc=foc)


def trace_ray_list_at_field(opt_model, ray_list, fld, wvl, foc, **kwargs):
    """rayoptics/raytr/trace.py:478-486: a list of ray DataFrames for the pupil points of
    ``ray_list`` at ``fld`` -- one ``trace_ray`` per point in the reference (``trace_field`` /
    ``trace_all_fields``: the boundary rays of every field), one launch per field here; the
    frames are built by the reference's own ``ray_df`` from the same segments (a failed ray
    contributes its partial packet, as ``rayerr_filter='full'`` does there)"""
    import rayoptics.raytr.trace as rtrace
    kw = dict(kwargs)
    output_filter = kw.pop('output_filter', None)
    rayerr_filter = kw.pop('rayerr_filter', 'full')          # trace_ray's default (trace.py:104)
    named = kw.get('use_named_tuples', False)
    kw['apply_vignetting'] = kw.get('apply_vignetting', True)
    pts = [np.asarray(p, dtype=float) for p in ray_list]
    px = np.array([p[0] for p in pts])
    py = np.array([p[1] for p in pts])
    pk = _trace_pupil(opt_model, fld, wvl, kw, output_filter, rayerr_filter, pupil_list=(px, py))
    ifcs = opt_model['seq_model'].ifcs
    rayset = []
    for r in range(len(pts)):
        ray_pkg, _err = emit(pk, r, output_filter, rayerr_filter, named, ifcs)
        ray, _op, _wvl = ray_pkg                             # (None here raises as there)
        rayset.append(ray.to_list() if hasattr(ray, 'to_list') else ray)
    return [rtrace.ray_df(r) for r in rayset]


def trace_astigmatism(opt_model, fld, wvl, foc, dx=0.001, dy=0.001):
    """rayoptics/raytr/trace.py:823-863 ``trace_astigmatism``: the sagittal and tangential focus
    shifts at ``fld`` from the chief ray's four close neighbours -- five ``trace_ray`` calls in
    the reference (``trace_astigmatism_curve``, :789-820, makes them for each of 21 field
    points: ``AstigmatismCurvePlot``), one five-ray launch here; the two line intersections are
    the reference's own ``intersect_2_lines`` on the same last segments.  A ray that fails
    sends the call to the reference's function, which then does whatever it does."""
    import rayoptics.raytr.trace as rtrace
    kwargs = {'apply_vignetting': True}                      # trace_base's default (trace.py:253)
    px = np.array([0., dx, 0., -dx, 0.])
    py = np.array([0., 0., dy, 0., -dy])
    pk = _trace_pupil(opt_model, fld, wvl, kwargs, 'last', None, pupil_list=(px, py),
                      out_mode=abi.OUT_LAST)
    if (pk.status[:5] != abi.OK).any():
        from . import install
        theirs = install._saved.get((rtrace, 'trace_astigmatism'))
        if theirs is not None:
            return theirs(opt_model, fld, wvl, foc, dx=dx, dy=dy)
        raise pk.error(int(np.argmax(pk.status[:5] != abi.OK)), opt_model['seq_model'].ifcs,
                       with_pkg=False)
    last = [pk.pkg(r)[0][-1] for r in range(5)]              # ray[-1] = [p, d, dst, nrml]
    s = rtrace.intersect_2_lines(last[1][0], last[1][1], last[3][0], last[3][1])
    s_foc = s * last[1][1][2]
    t = rtrace.intersect_2_lines(last[2][0], last[2][1], last[4][0], last[4][1])
    t_foc = t * last[2][1][2]
    if foc is not None:
        focus_shift = foc
        s_foc -= focus_shift
        t_foc -= focus_shift
    return s_foc, t_foc


def _launch_setup(opt_model, fld, wvl, kwargs, out_mode, foc=0.0, image_pt=(0., 0.), wf=None):
    """engine, field constants, wavelength index and options of one pupil launch:
    trace_base's preamble (rayoptics/raytr/trace.py:253-310).  Every branch of
    ray_start_from_osp runs on the device (``rox_field.kind``)."""
    pupil_type = kwargs.get('pupil_type', 'rel pupil')
    eng = session.engine_for(opt_model)
    tbl = eng.table
    opts = opts_from_kwargs(tbl.n_ifcs, kwargs, out_mode, foc, image_pt, wf)
    f = field_from_model(opt_model, fld, pupil_type, cache=eng.memo.obj_coords)
    if pupil_type != 'rel pupil':
        opts.flags &= ~abi.APPLY_VIGNETTING             # trace.py:291-295
    if f.kind == abi.FLD_EPD_WIDE or f.z_dir0 == 0.0:
        opts.flags &= ~abi.INTERSECT_OBJ                # trace.py:302-303 (any wide-angle field)
    return eng, f, tbl.wvl_index(wvl), opts


def _trace_pupil(opt_model, fld, wvl, kwargs, output_filter, rayerr_filter, grid=None,
                 pupil_list=None,
                 out_mode=None, foc=0.0, image_pt=(0., 0.), wf=None):
    if out_mode is None:
        # partial packets (rayerr_filter='full') need the FULL layout
        out_mode = (abi.OUT_LAST if output_filter == 'last' and rayerr_filter != 'full'
                    else abi.OUT_FULL)
    eng, f, wi, opts = _launch_setup(opt_model, fld, wvl, kwargs, out_mode, foc, image_pt, wf)
    # small launches (fans, rim rays, ray lists, figure-sized grids): NumPy buffers through the
    # library's host-pointer path -- one launch, one synchronise, no device-to-host copies
    from .engine import HOST_DIRECT_BYTES, grid_rays
    R = grid_rays(grid) if grid is not None else len(pupil_list[0])
    per_ray = 8 * (abi.SEG_DOUBLES * (eng.table.n_ifcs if out_mode == abi.OUT_FULL else 1) + 4)
    if R * per_ray <= HOST_DIRECT_BYTES and hasattr(eng, 'trace_pupil_np'):
        if grid is not None:
            h = eng.trace_pupil_np(f, wi, opts, grid=grid)
        else:
            h = eng.trace_pupil_np(f, wi, opts, px=pupil_list[0], py=pupil_list[1])
        return HostPackets(h, eng.table, opts.flags, out_mode, wvl)
    if grid is not None:
        res = eng.trace_pupil_grid(f, grid, wi, opts)
    else:
        res = eng.trace_pupil_list(f, pupil_list[0], pupil_list[1], wi, opts)
    return HostPackets(res.to_host(), eng.table, opts.flags, out_mode, wvl)


def trace_grid(opt_model, grid_rng, fld, wvl, foc, img_filter=None,
               form='grid', append_if_none=True, **kwargs):
    """rayoptics/raytr/trace.py:563-605"""
    output_filter = kwargs.pop('output_filter', None)
    rayerr_filter = kwargs.pop('rayerr_filter', None)
    named = kwargs.get('use_named_tuples', False)
    num = grid_rng[2]
    if 'check_apertures' in kwargs:     # the reference passes it itself (:581-583)
        raise TypeError("rayoptics.raytr.trace.trace_safe() got multiple values for keyword "
                        "argument 'check_apertures'")
    kwargs['check_apertures'] = True                                 # :583
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', True)   # trace_base default
    pk = _trace_pupil(opt_model, fld, wvl, kwargs, output_filter, rayerr_filter,
                      grid=make_grid(grid_rng[0], grid_rng[1], num))
    ifcs = opt_model['seq_model'].ifcs
    grid = []
    pup = np.ascontiguousarray(pk.pupil[:2].T)      # [R, 2]: one (x, y) row per ray
    for i in range(num):
        working_grid = grid if form == 'list' else []
        for j in range(num):
            r = i * num + j
            pupil = pup[r]
            pkg, _err = emit(pk, r, output_filter, rayerr_filter, named, ifcs)
            if pkg is not None:
                if img_filter:
                    working_grid.append(img_filter(pupil, pkg))
                else:
                    working_grid.append([pupil[0], pupil[1], pkg])
            else:
                if img_filter:
                    result = img_filter(pupil, None)
                    if result is not None or append_if_none:
                        working_grid.append(result)
                elif append_if_none:
                    working_grid.append([pupil[0], pupil[1], None])
        if form == 'grid':
            grid.append(working_grid)
    # :605 as the reference has it: ragged entries (packets, or an array-returning img_filter
    # beside None entries) make NumPy >= 1.24 raise ValueError here exactly as they do in the
    # reference; older NumPy builds the object array with its deprecation warning in both
    return np.array(grid)


def trace_fan(opt_model, fan_rng, fld, wvl, foc, img_filter=None, **kwargs):
    """rayoptics/raytr/trace.py:537-560"""
    output_filter = kwargs.pop('output_filter', None)
    rayerr_filter = kwargs.pop('rayerr_filter', None)
    named = kwargs.get('use_named_tuples', False)
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', True)
    pk = _trace_pupil(opt_model, fld, wvl, kwargs, output_filter, rayerr_filter,
                      grid=make_grid(fan_rng[0], fan_rng[1], fan_rng[2], abi.GRID_FAN))
    ifcs = opt_model['seq_model'].ifcs
    fan = []
    pup = np.ascontiguousarray(pk.pupil[:2].T)
    for r in range(fan_rng[2]):
        pupil = pup[r]
        pkg, _err = emit(pk, r, output_filter, rayerr_filter, named, ifcs)
        if pkg is not None:
            fan.append([pupil, img_filter(pupil, pkg) if img_filter else pkg])
    return fan


def trace_grid_spot(opt_model, grid_rng, fld, wvl, foc, image_pt, **kwargs):
    """fused spot diagram: the transverse aberrations SpotDiagramFigure's
    ``spot`` filter computes (rayoptics/mpl/axisarrayfigure.py:229-238), for the
    rays that get through, as one (R_ok, 2) array -- what
    ``seq_model.trace_grid(spot, fi, wl, num_rays, form='list',
    append_if_none=False)`` returns per wavelength (sequential.py:1058-1085);
    ROX_OUT_HITS_COMPACT output mode."""
    kwargs['check_apertures'] = True
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', True)
    eng, f, wi, opts = _launch_setup(opt_model, fld, wvl, kwargs, abi.OUT_HITS_COMPACT,
                                     foc, image_pt[:2])
    # survivors are packed in ray order on the device and written straight into
    # pinned host memory: the array returned *is* that buffer
    return eng.trace_pupil_grid_hits(f, make_grid(grid_rng[0], grid_rng[1], grid_rng[2]),
                                     wi, opts)


def trace_grid_spot_stats(opt_model, grid_rng, fld, wvl, foc, image_pt, bins=None, **kwargs):
    """What the consumers of a spot diagram reduce it to, without the spot leaving the device:
    ``(summary, hist, x_edges, y_edges)`` of the transverse aberrations of the pupil grid --
    ``summary``: n, centroid, rms_radius, min / max (RayGeoPSF.ray_data_bounds,
    rayoptics/mpl/analysisfigure.py:237-248); ``hist``: ``numpy.histogram2d(x, y, bins=[x_edges,
    y_edges])[0]`` as RayGeoPSF.plot's ``hist2d`` forms it (:250-290).  ``bins``: None (no
    histogram), ``(x_edges, y_edges)``, or an int ``num`` -- RayGeoPSF's 'fit' scale: edges =
    ``linspace`` over the larger half-extent of the data about (0, centre_y), ``num`` samples
    per axis, made from the summary of a first pass (one launch, two tiny reductions).  A
    ROX_OUT_HITS launch + rox_spot_stats: 72 bytes and the histogram cross PCIe."""
    from .engine import DeviceResult
    kwargs['check_apertures'] = True
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', True)
    eng, f, wi, opts = _launch_setup(opt_model, fld, wvl, kwargs, abi.OUT_HITS, foc, image_pt[:2])
    grid = make_grid(grid_rng[0], grid_rng[1], grid_rng[2])
    R = grid_rng[2] * grid_rng[2]
    # (the launch's rows stay on the device: one buffer per thread and size, kept on the engine)
    key = (threading.get_ident(), R)
    res = eng.memo.scratch.get(key)
    if res is None:
        if len(eng.memo.scratch) >= 8:
            eng.memo.scratch.clear()
        res = eng.memo.scratch[key] = DeviceResult(eng.torch, eng.device, 0, R, abi.OUT_HITS,
                                                   want_pupil=False, nan_fill=False)
    eng.trace_pupil_grid(f, grid, wi, opts, want_pupil=False, out=res)
    if bins is None:
        summ, _ = eng.spot_stats(res)
        return summ, None, None, None
    if isinstance(bins, int):
        summ, _ = eng.spot_stats(res)
        # analysisfigure.py:237-262 ('fit'): delta = (max - min) / 2, centre_y = (max_y + min_y) / 2
        dx = (summ['max'][0] - summ['min'][0]) / 2
        dy = (summ['max'][1] - summ['min'][1]) / 2
        cy = (summ['max'][1] + summ['min'][1]) / 2
        mv = dx if dx > dy else dy
        x_edges = np.linspace(-mv, mv, num=bins)
        y_edges = np.linspace(cy - mv, cy + mv, num=bins)
    else:
        x_edges, y_edges = bins
    summ, hist = eng.spot_stats(res, x_edges, y_edges)
    return summ, hist, np.asarray(x_edges), np.asarray(y_edges)


def trace_grid_spots(opt_model, grid_rng, fld, wvls, foc, image_pt, **kwargs):
    """``trace_grid_spot`` for every wavelength of ``wvls`` in ONE launch
    (rox_trace_pupil_grids): the per-wavelength loop of SequentialModel.trace_grid
    (sequential.py:1073-1084) -- a list of (R_ok, 2) arrays, one per wavelength."""
    kwargs['check_apertures'] = True
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', True)
    flds, wis, optl = [], [], []
    eng = None
    for wvl in wvls:
        eng, f, wi, opts = _launch_setup(opt_model, fld, wvl, dict(kwargs), abi.OUT_HITS_COMPACT,
                                         foc, image_pt[:2])
        flds.append(f)
        wis.append(wi)
        optl.append(opts)
    return eng.trace_pupil_grids_hits(flds, wis, make_grid(grid_rng[0], grid_rng[1], grid_rng[2]), optl)


def _is_spot_filter(fct):
    """SpotDiagramFigure's own callback (rayoptics/mpl/axisarrayfigure.py:229-238):
    a closure, so it can only be recognised by where it was defined -- module and
    qualified name, and the four-variable closure (self is not among them) it has
    in the reference"""
    return (getattr(fct, '__module__', '') == 'rayoptics.mpl.axisarrayfigure'
            and getattr(fct, '__qualname__', '') == 'SpotDiagramFigure.__init__.<locals>.spot')


def seq_trace_grid(self, fct, fi, wl=None, num_rays=21, form='grid',
                   append_if_none=True, **kwargs):
    with session.hold(self.opt_model):      # one validation of the model for the whole call
        return _seq_trace_grid(self, fct, fi, wl, num_rays, form, append_if_none, **kwargs)


def _seq_trace_grid(self, fct, fi, wl=None, num_rays=21, form='grid',
                    append_if_none=True, **kwargs):
    """rayoptics/seq/sequential.py:1058-1085, as a replacement *method* of
    SequentialModel.  Chief-ray / reference-sphere setup stays the reference's
    (a handful of iterated single rays); the num_rays**2 loop goes to the GPU.
    SpotDiagramFigure's own ``spot`` callback is recognised and fused."""
    from rayoptics.raytr import trace as ref_trace
    osp = self.opt_model.optical_spec
    wvls = osp.spectral_region
    wvl = self.central_wavelength()
    wv_list = wvls.wavelengths if wl is None else [wl]
    fld = osp.field_of_view.fields[fi]
    foc = osp.defocus.get_focus()

    rs_pkg, cr_pkg = ref_trace.setup_pupil_coords(self.opt_model, fld, wvl, foc)
    fld.chief_ray = cr_pkg
    fld.ref_sphere = rs_pkg

    grids = []
    grid_def = [np.array([-1., -1.]), np.array([1., 1.]), num_rays]
    fused = _is_spot_filter(fct) and form == 'list' and not append_if_none
    if fused and len(wv_list) > 1:
        # every wavelength of the field in one launch
        return (trace_grid_spots(self.opt_model, grid_def, fld, list(wv_list), foc,
                                 fld.ref_sphere[0], **dict(kwargs)), wvls.render_colors)
    for wi, wvl in enumerate(wv_list):
        if fused:
            grid = trace_grid_spot(self.opt_model, grid_def, fld, wvl, foc,
                                   fld.ref_sphere[0], **dict(kwargs))
        else:
            grid = trace_grid(self.opt_model, grid_def, fld, wvl, foc,
                              form=form, append_if_none=append_if_none,
                              img_filter=lambda p, ray_pkg, wi=wi, wvl=wvl:
                              fct(p, wi, ray_pkg, fld, wvl, foc),
                              **dict(kwargs))
        grids.append(grid)
    return grids, wvls.render_colors


_FAN_FIGURE_CALLBACKS = {'RayFanFigure.__init__.<locals>.ray_abr': 'ray',
                         'RayFanFigure.__init__.<locals>.opd': 'opd'}


def _fan_figure_callback(fct):
    """RayFanFigure's own two callbacks (rayoptics/mpl/axisarrayfigure.py:110-129):
    closures, recognised -- like SpotDiagramFigure's `spot` -- by module and qualified name"""
    if getattr(fct, '__module__', '') != 'rayoptics.mpl.axisarrayfigure':
        return None
    return _FAN_FIGURE_CALLBACKS.get(getattr(fct, '__qualname__', ''))


def seq_trace_fan(self, fct, fi, xy, num_rays=21, **kwargs):
    with session.hold(self.opt_model):      # one validation of the model for the whole call
        return _seq_trace_fan(self, fct, fi, xy, num_rays, **kwargs)


def _seq_trace_fan(self, fct, fi, xy, num_rays=21, **kwargs):
    """rayoptics/seq/sequential.py:1006-1056, as a replacement *method* of SequentialModel:
    the x or y fan of field ``fi`` at every wavelength.  Chief ray and reference sphere per
    wavelength stay the reference's; for RayFanFigure's own callbacks (transverse aberration,
    OPD) the fans of all wavelengths are ONE launch in ROX_OUT_FAN mode (per ray: dx, dy and
    the OPD against that wavelength's reference sphere) instead of a packet per ray and a
    Python callback on each."""
    from rayoptics.raytr import trace as ref_trace
    from .table import wavefront_from_model, UnsupportedModelError
    opt_model = self.opt_model
    osp = opt_model.optical_spec
    fld = osp.field_of_view.fields[fi]
    wvl = self.central_wavelength()
    foc = osp.defocus.get_focus()

    rs_pkg, cr_pkg = ref_trace.setup_pupil_coords(opt_model, fld, wvl, foc)
    fld.chief_ray = cr_pkg
    fld.ref_sphere = rs_pkg
    # the central wavelength's image point is the reference for every wavelength (:1019-1020)
    ref_img_pt = rs_pkg[0]

    wvls = osp.spectral_region
    fan_start = np.array([0., 0.])
    fan_stop = np.array([0., 0.])
    fan_start[xy] = -1.0
    fan_stop[xy] = 1.0
    fan_def = [fan_start, fan_stop, num_rays]
    kind = _fan_figure_callback(fct)
    fusable = (kind is not None and kwargs.get('output_filter') is None
               and kwargs.get('rayerr_filter') is None and not kwargs.get('filter_out_phantoms', False))
    rc, setups = [], []
    for wi, w in enumerate(wvls.wavelengths):
        rc.append(wvls.render_colors[wi])
        rs_pkg, cr_pkg = ref_trace.setup_pupil_coords(opt_model, fld, w, foc, image_pt=ref_img_pt)
        fld.chief_ray = cr_pkg
        fld.ref_sphere = rs_pkg
        wf = None
        if fusable:
            try:
                wf = wavefront_from_model(opt_model, fld, cr_pkg, rs_pkg)
            except UnsupportedModelError:
                fusable = False
        setups.append((w, rs_pkg, cr_pkg, wf))

    fans = []                      # per wavelength: [(pupil[xy], value)]
    if fusable:
        kw = dict(kwargs)
        kw['apply_vignetting'] = kw.get('apply_vignetting', True)
        flds, wis, optl = [], [], []
        eng = None
        for w, rs_pkg, _cr, wf in setups:
            eng, f, widx, opts = _launch_setup(opt_model, fld, w, dict(kw), abi.OUT_FAN, foc,
                                               rs_pkg[0][:2], wf)
            flds.append(f)
            wis.append(widx)
            optl.append(opts)
        # (every wavelength's fan in one launch, written by the kernel straight into pinned host
        # memory: one synchronise instead of three device-to-host copies per wavelength)
        res = eng.trace_pupil_grids_host(flds, wis, make_grid(fan_def[0], fan_def[1], num_rays, abi.GRID_FAN),
                                         optl)
        for (w, _rs, _cr, _wf), h in zip(setups, res):
            seg = h.seg if h.seg.ndim == 2 else h.seg[0]       # [3][num_rays]: dx, dy, OPD
            if kind == 'opd':
                # convert_to_waves = 1/self.wvl_to_sys_units(wvl) of the figure (:126)
                val = (1 / opt_model.nm_to_sys_units(w)) * seg[2]
            else:
                val = seg[xy]
            fans.append([(h.pupil[xy, k], val[k]) for k in range(num_rays) if h.status[k] == abi.OK])
    else:
        for w, rs_pkg, cr_pkg, _wf in setups:
            fld.chief_ray = cr_pkg
            fld.ref_sphere = rs_pkg
            fan = trace_fan(opt_model, fan_def, fld, w, foc,
                            img_filter=lambda p, ray_pkg, w=w: fct(p, xy, ray_pkg, fld, w, foc),
                            **dict(kwargs))
            fans.append([(p[xy], y_val) for p, y_val in fan])
    # (the field is left with the last wavelength's chief ray and sphere, as in the reference)
    fans_x, fans_y = [], []
    max_rho_val = 0.0
    max_y_val = 0.0
    for fan in fans:
        f_x, f_y = [], []
        for x_val, y_val in fan:
            f_x.append(x_val)
            f_y.append(y_val)
            if abs(x_val) > max_rho_val:
                max_rho_val = abs(x_val)
            if abs(y_val) > max_y_val:
                max_y_val = abs(y_val)
        fans_x.append(f_x)
        fans_y.append(f_y)
    return np.array(fans_x), np.array(fans_y), (max_rho_val, max_y_val), rc


# ---- chief-ray aiming --------------------------------------------------------
def _aim_problem(opt_model, fld, wvl, tbl, stop):
    """rox_aim for iterate_ray (trace.py:313-415) aiming at the centre of the stop: the 1-D
    branch for a field on the y axis (:376-392), the 2-D branch otherwise (:394-410, fsolve
    with epsfcn = 0.0001 * fod.enp_radius).  None for wide-angle models, whose pupil search
    is wideangle.find_real_enp (trace.py:634-635)."""
    osp = opt_model['optical_spec']
    if osp['fov'].is_wide_angle:
        return None
    fod = opt_model['analysis_results']['parax_data'].fod
    pt0, _d0 = osp.obj_coords(fld)
    a = abi.Aim()
    for i in range(3):
        a.pt0[i] = float(pt0[i])
    a.z_enp = float(fod.obj_dist + fod.enp_dist)
    a.x_target = a.y_target = 0.0
    a.z_dir0 = float(opt_model['seq_model'].z_dir[0])
    a.wvl_idx = tbl.wvl_index(wvl)
    a.surf = int(stop)
    a.flip = 1
    a.two_d = 0 if pt0[0] == 0.0 else 1
    a.epsfcn = float(0.0001 * fod.enp_radius)
    return a


def _enp_problem(opt_model, fld, wvl, tbl, stop):
    """rox_enp for wideangle.find_real_enp (wideangle.py:96-134): the field's object-space
    direction, the rotation enp_z_coordinate applies to the pupil point, the paraxial pupil"""
    from rayoptics.util.misc_math import rot_v1_into_v2
    osp = opt_model['optical_spec']
    fod = opt_model['analysis_results']['parax_data'].fod
    _pt0, dir0 = osp.obj_coords(fld)
    rot = rot_v1_into_v2(np.array([0., 0., 1.]), dir0)
    e = abi.Enp()
    for i in range(3):
        e.dir0[i] = float(dir0[i])
        for j in range(3):
            e.rot[3 * i + j] = float(rot[i][j])
    from .table import rt_order_of
    e.rot_order = rt_order_of(rot)
    e.obj_dist = float(fod.obj_dist)
    e.z_enp_0 = float(fod.enp_dist)
    e.aim_info = float('nan') if fld.aim_info is None else float(fld.aim_info)
    e.wvl_idx = tbl.wvl_index(wvl)
    e.surf = 1 if stop is None else int(stop)
    e.check_direction = 1
    return e


def find_real_enps(opt_model, stop_idx, flds, wvl):
    """wideangle.find_real_enp for every field of ``flds`` in one launch (one lane each):
    [(z_enp, z of the last trial ray, abi.ENP_*)]"""
    eng = session.engine_for(opt_model)
    out = [(0.0, 0.0, abi.ENP_REFERENCE_RAISES)] * len(flds)
    # (an aim_info that is not a scalar -- left over from a non-wide-angle aim -- is the
    # reference's own business: those fields are routed to its function)
    where = [k for k, fld in enumerate(flds) if np.ndim(fld.aim_info) == 0]
    if where:
        probs = [_enp_problem(opt_model, flds[k], wvl, eng.table, stop_idx) for k in where]
        z, result = eng.find_real_enp(probs)
        for j, k in enumerate(where):
            out[k] = (float(z[j, 0]), float(z[j, 1]), int(result[j]))
    return out


def find_real_enp(opm, stop_idx, fld, wvl, vselector='rev1'):
    """rayoptics/raytr/wideangle.py:86-94 (the 'rev1' search, :96-292): (z_enp, rr) with
    the whole search -- sampled walk, find_edge, newton, brentq -- run by one device lane;
    ``rr`` is the reference's RayResult of the last trial ray, retraced once."""
    import rayoptics.raytr.wideangle as wa
    ref_fn = getattr(wa.find_real_enp, '__wrapped__', wa.find_real_enp)
    if vselector != 'rev1':
        return ref_fn(opm, stop_idx, fld, wvl, vselector=vselector)
    (z_enp, z_last, code), = find_real_enps(opm, stop_idx, [fld], wvl)
    if code == abi.ENP_REFERENCE_RAISES:
        return ref_fn(opm, stop_idx, fld, wvl)      # raises what the reference raises
    osp = opm['optical_spec']
    fod = opm['analysis_results']['parax_data'].fod
    _pt0, dir0 = osp.obj_coords(fld)
    _coord, rr = wa.enp_z_coordinate(z_last, opm['seq_model'], 1 if stop_idx is None else stop_idx,
                                     dir0, fod.obj_dist, wvl)
    return z_enp, rr


def aim_chief_rays(opt_model, flds, wvl=None):
    """aim_info for every field in ``flds``, all solved together in one launch (one lane
    each): fields on the y axis by the secant iteration of scipy.optimize.newton, the others
    by MINPACK's hybrd as scipy.optimize.fsolve runs it -- both restated on the device
    (rayoptics/raytr/trace.py:313-415).  Wide-angle models: the pupil search of
    wideangle.find_real_enp (wideangle.py:86-427), one lane per field (rox_find_real_enp)."""
    from rayoptics.raytr import trace as ref_trace
    sm = opt_model['seq_model']
    if wvl is None:
        wvl = sm.central_wavelength()
    stop = sm.stop_surface
    out = [None] * len(flds)
    if stop is None and not opt_model['optical_spec']['fov'].is_wide_angle:
        return [np.array([0., 0.]) + np.array([0., 0.]) for _ in flds]      # floating stop, :412-413
    if opt_model['optical_spec']['fov'].is_wide_angle:          # trace.py:634-635
        for k, (z_enp, _z_last, code) in enumerate(find_real_enps(opt_model, stop, flds, wvl)):
            if code == abi.ENP_REFERENCE_RAISES:
                out[k] = _ref_aim_chief_ray(ref_trace)(opt_model, flds[k], wvl)
            else:
                out[k] = z_enp
        return out
    eng = session.engine_for(opt_model)
    probs, where = [], []
    for k, fld in enumerate(flds):
        a = _aim_problem(opt_model, fld, wvl, eng.table, stop) if stop is not None else None
        if a is None:
            out[k] = _ref_aim_chief_ray(ref_trace)(opt_model, fld, wvl)
        else:
            probs.append(a)
            where.append(k)
    if probs:
        aim, _result = eng.aim_chief_rays(probs)
        for k, xy in zip(where, aim):
            out[k] = np.array([float(xy[0]), float(xy[1])])
    return out


def _ref_aim_chief_ray(ref_trace):
    """the reference's own aim_chief_ray, also while install() has it rebound"""
    fn = ref_trace.aim_chief_ray
    return getattr(fn, '__wrapped__', fn)


def aim_chief_ray(opt_model, fld, wvl=None):
    """rayoptics/raytr/trace.py:627-640"""
    return aim_chief_rays(opt_model, [fld], wvl)[0]


def osp_update_optical_properties(self, **kwargs):
    """rayoptics/raytr/opticalspec.py:263-281 as a replacement *method* of
    OpticalSpecs: first-order data from the reference, then every field's chief
    ray aimed in one device launch instead of a Python loop of iterated rays."""
    from rayoptics.parax.firstorder import compute_first_order
    opm = self.opt_model
    sm = opm['seq_model']
    if sm.get_num_surfaces() > 2:
        src = kwargs.get('src_model', None)
        stop = sm.stop_surface
        wvl = self.spectral_region.central_wvl
        opm['analysis_results']['parax_data'] = compute_first_order(opm, stop, wvl, src_model=src)
        if self.do_aiming:
            flds = self.field_of_view.fields
            try:
                for fld, aim in zip(flds, aim_chief_rays(opm, flds, wvl)):
                    fld.aim_info = aim
            except Exception:           # per-field failures are reported, not raised (:277-281)
                for i, fld in enumerate(flds):
                    try:
                        fld.aim_info = aim_chief_ray(opm, fld, wvl)
                    except Exception:
                        print(f"OpticalSpecs aim_chief_ray failure at field {i}")

"""Drop-ins for the list / grid / fan drivers of ``rayoptics.raytr.analyses``.

  trace_list_of_rays <- rayoptics/raytr/analyses.py:458-510   (+ trace_rays_soa, its array form)
  trace_ray_list     <- rayoptics/raytr/analyses.py:437-455
  trace_ray_grid     <- rayoptics/raytr/analyses.py:666-696
  trace_ray_fan      <- rayoptics/raytr/analyses.py:212-230
  eval_fan / trace_fan / focus_fan <- rayoptics/raytr/analyses.py:233-345 (RayFan: dx, dy
                        and OPD of every fan ray in one launch, ROX_OUT_FAN)
  eval_wavefront     <- rayoptics/raytr/analyses.py:699-732 (OPD fused on device)
  trace_wavefront / focus_wavefront <- rayoptics/raytr/analyses.py:735-791 (RayGrid, PSF)
  trace_pupil_coords / focus_pupil_coords <- rayoptics/raytr/analyses.py:545-580 (RayList, RayGeoPSF)
"""
import numpy as np

from . import abi, session
from .engine import make_grid
from .raypkg import HostPackets
from .trace import opts_from_kwargs, emit, _trace_pupil, _launch_setup


def _setup_pupil_coords(opt_model, fld, wvl, foc, image_pt=None, image_delta=None):
    """trace.setup_pupil_coords (rayoptics/raytr/trace.py:608-624): the reference's
    own for a live model (chief ray cached on the field, reference sphere host
    math); table-backed models bring theirs"""
    own = getattr(opt_model, 'setup_pupil_coords', None)
    if own is not None:
        return own(fld, wvl, foc, image_pt=image_pt, image_delta=image_delta)
    from rayoptics.raytr import trace as ref_trace
    return ref_trace.setup_pupil_coords(opt_model, fld, wvl, foc, image_pt=image_pt,
                                        image_delta=image_delta)


def trace_list_of_rays(opt_model, rays, output_filter=None, rayerr_filter=None,
                       **kwargs):
    """explicit (pt0, dir0, wvl) rays; per-ray wavelengths allowed"""
    rays = list(rays)
    eng = session.engine_for(opt_model)
    tbl = eng.table
    R = len(rays)
    pt0 = np.empty((3, R))
    dir0 = np.empty((3, R))
    wvls = np.empty(R)
    wi = np.empty(R, dtype=np.int32)
    for r, (p, d, w) in enumerate(rays):
        pt0[:, r], dir0[:, r], wvls[r] = p, d, w
        wi[r] = tbl.wvl_index(w)
    # partial packets (rayerr_filter='full') need the FULL layout
    out_mode = (abi.OUT_LAST if output_filter == 'last' and rayerr_filter != 'full'
                else abi.OUT_FULL)
    opts = opts_from_kwargs(tbl.n_ifcs, kwargs, out_mode)
    from .engine import HOST_DIRECT_BYTES
    per_ray = 8 * (abi.SEG_DOUBLES * (tbl.n_ifcs if out_mode == abi.OUT_FULL else 1) + 8)
    if R * per_ray <= HOST_DIRECT_BYTES and hasattr(eng, 'trace_rays_np'):
        host = eng.trace_rays_np(pt0, dir0, wi, opts)       # (small lists: no device-to-host copies)
    else:
        host = eng.trace_rays(pt0, dir0, wi, opts).to_host()
    pk = HostPackets(host, tbl, opts.flags, out_mode, wvls)
    ifcs = opt_model['seq_model'].ifcs
    named = True        # trace.trace() wraps in RayPkg (trace.py:250)
    ray_list = []
    for r, ray in enumerate(rays):
        if pk.status[r] != abi.OK:
            if rayerr_filter == 'full':
                ray_list.append((ray, pk.error(r, ifcs, with_pkg=True, named=False)))
            elif rayerr_filter == 'summary':
                ray_list.append((ray, pk.error(r, ifcs, with_pkg=False)))
            continue
        pkg = pk.pkg(r, named)
        if output_filter is None:
            ray_list.append(pkg)
        elif output_filter == 'last':
            seg, op_delta, wvl = pkg
            ray_list.append((seg[-1], op_delta, wvl))
        else:
            ray_list.append(output_filter(pkg))
    return ray_list


def trace_rays_soa(opt_model, pt0, dir0, wvl, out_mode=abi.OUT_FULL, on_device=False, **kwargs):
    """array form of :func:`trace_list_of_rays` for large batches: ``pt0``,
    ``dir0`` are ``[3, R]`` arrays (numpy or torch), ``wvl`` one wavelength or
    ``R`` of them (nm, members of the spectral region).  No per-ray Python
    objects are created: returns the engine's ``DeviceResult`` (``on_device``)
    or a :class:`~.raypkg.HostPackets`, whose ``pkg(r)`` / ``error(r)`` give the
    reference-shaped view of any single ray on demand."""
    eng = session.engine_for(opt_model)
    tbl = eng.table
    if np.ndim(wvl) == 0:
        wi = tbl.wvl_index(wvl)
    else:
        lut = {w: i for i, w in enumerate(tbl.wvls)}
        wi = np.fromiter((lut[float(w)] for w in wvl), dtype=np.int32, count=len(wvl))
    opts = opts_from_kwargs(tbl.n_ifcs, kwargs, out_mode)
    res = eng.trace_rays(pt0, dir0, wi, opts)
    if on_device:
        return res
    return HostPackets(res.to_host(), tbl, opts.flags, out_mode, wvl)


def trace_ray_list(opt_model, pupil_coords, fld, wvl, foc, append_if_none=False,
                   output_filter=None, rayerr_filter=None, **kwargs):
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', True)
    named = kwargs.get('use_named_tuples', False)
    items = pupil_coords if isinstance(pupil_coords, (list, np.ndarray)) else list(pupil_coords)
    if isinstance(items, np.ndarray) and items.ndim == 2 and items.shape[1] >= 2 and items.dtype.kind == 'f':
        pc = items[:, :2].astype(float)
    else:
        pc = np.array([[p[0], p[1]] for p in items], dtype=float).reshape(-1, 2)
    pk = _trace_pupil(opt_model, fld, wvl, kwargs, output_filter, rayerr_filter,
                      pupil_list=(pc[:, 0].copy(), pc[:, 1].copy()))
    ifcs = opt_model['seq_model'].ifcs
    ray_list = []
    # Field.apply_vignetting scales `pupil[:]`: a view for an ndarray (the caller's
    # coordinates change in place), a copy for a list or tuple
    whole = isinstance(items, np.ndarray) and items.ndim == 2 and items.shape[1] >= 2
    if whole:                                   # every row at once
        items[:, 0], items[:, 1] = pk.pupil[0], pk.pupil[1]
    for r, p in enumerate(items):
        if not whole and isinstance(p, np.ndarray):
            p[0], p[1] = pk.pupil[0, r], pk.pupil[1, r]
        pkg, _err = emit(pk, r, output_filter, rayerr_filter, named, ifcs)
        if pkg is not None:
            ray_list.append([p[0], p[1], pkg])
        elif append_if_none:
            ray_list.append([p[0], p[1], None])
    return ray_list


def trace_ray_grid(opt_model, grid_rng, fld, wvl, foc, append_if_none=True,
                   output_filter=None, rayerr_filter=None, **kwargs):
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', False)   # :674
    named = kwargs.get('use_named_tuples', False)
    num = grid_rng[2]
    pk = _trace_pupil(opt_model, fld, wvl, kwargs, output_filter, rayerr_filter,
                      grid=make_grid(grid_rng[0], grid_rng[1], num))
    ifcs = opt_model['seq_model'].ifcs
    grid = []
    for i in range(num):
        row = []
        for j in range(num):
            r = i * num + j
            pkg, _err = emit(pk, r, output_filter, rayerr_filter, named, ifcs)
            if pkg is not None:
                row.append([pk.pupil[0, r], pk.pupil[1, r], pkg])
            elif append_if_none:
                row.append([pk.pupil[0, r], pk.pupil[1, r], None])
        grid.append(row)
    return grid


def trace_ray_fan(opt_model, fan_rng, fld, wvl, foc, output_filter=None,
                  rayerr_filter=None, **kwargs):
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', True)
    pk = _trace_pupil(opt_model, fld, wvl, kwargs, output_filter, rayerr_filter,
                      grid=make_grid(fan_rng[0], fan_rng[1], fan_rng[2], abi.GRID_FAN))
    ifcs = opt_model['seq_model'].ifcs
    fan = []
    for r in range(fan_rng[2]):
        pkg, _err = emit(pk, r, output_filter, rayerr_filter, True, ifcs)   # :222
        if pkg is not None:
            fan.append([pk.pupil[0, r], pk.pupil[1, r], pkg])
    return fan


def eval_wavefront(opt_model, fld, wvl, foc, image_pt_2d=None, image_delta=None,
                   num_rays=21, value_if_none=np.nan, **kwargs):
    """rayoptics/raytr/analyses.py:699-732: OPD (in waves) over the vignetted
    pupil bounding box.  Chief ray and reference sphere come from the
    reference's own ``setup_pupil_coords``; the num_rays**2 traces *and* their
    ``wave_abr_full_calc`` (rayoptics/raytr/waveabr.py:256-307) run in one
    launch (ROX_OUT_OPD).  Packet filters or an infinite reference sphere take
    the generic route: device trace, reference ``waveabr`` on the lazy views."""
    from .table import wavefront_from_model, UnsupportedModelError
    ref_sphere, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, foc, image_pt_2d, image_delta)
    fld.chief_ray = cr_pkg
    fld.ref_sphere = ref_sphere
    oversize = kwargs.get('oversize', 1.)
    vig_bbox = fld.vignetting_bbox(opt_model['osp']['pupil'], oversize=oversize)
    grid_def = [vig_bbox[0], vig_bbox[1], num_rays]
    kwargs['check_apertures'] = kwargs.get('check_apertures', True)
    convert_to_opd = 1 / opt_model.nm_to_sys_units(wvl)
    fused = kwargs.get('output_filter') is None and kwargs.get('rayerr_filter') is None \
        and not kwargs.get('filter_out_phantoms', False)
    wf = None
    if fused:
        try:
            wf = wavefront_from_model(opt_model, fld)
        except UnsupportedModelError:
            fused = False
    if not fused:
        from rayoptics.raytr import waveabr
        fod = opt_model['analysis_results']['parax_data'].fod
        grid = trace_ray_grid(opt_model, grid_def, fld, wvl, foc, **kwargs)
        return np.array([[(px, py, convert_to_opd * waveabr.wave_abr_full_calc(
            fod, fld, wvl, foc, pkg, cr_pkg, ref_sphere)) if pkg is not None
            else (px, py, value_if_none) for px, py, pkg in row] for row in grid])
    kwargs.pop('output_filter', None)
    kwargs.pop('rayerr_filter', None)
    kwargs['apply_vignetting'] = kwargs.get('apply_vignetting', False)     # trace_ray_grid :674
    pk = _trace_pupil(opt_model, fld, wvl, kwargs, None, None,
                      grid=make_grid(grid_def[0], grid_def[1], num_rays),
                      out_mode=abi.OUT_OPD, wf=wf)
    ok = pk.status == abi.OK
    opd = np.where(ok, convert_to_opd * pk.seg[0, 0], value_if_none)
    out = np.stack([pk.pupil[0], pk.pupil[1], opd], axis=1)
    return out.reshape(num_rays, num_rays, 3)


def seq_trace_wavefront(self, fld, wvl, foc, num_rays=32):
    with session.hold(self.opt_model):      # one validation of the model for the whole call
        return _seq_trace_wavefront(self, fld, wvl, foc, num_rays)


def _seq_trace_wavefront(self, fld, wvl, foc, num_rays=32):
    """rayoptics/seq/sequential.py:1087-1114 as a replacement *method* of
    SequentialModel: [x, y, opd] over the unit pupil square, opd in waves
    (``opd / nm_to_sys_units(wvl)``), 0.0 where the ray failed.  The reference
    reaches this through trace.trace_grid with a per-ray ``wave_abr_full_calc``
    callback; here trace and OPD are one ROX_OUT_OPD launch."""
    from .table import wavefront_from_model
    opt_model = self.opt_model
    rs_pkg, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, foc)
    fld.chief_ray = cr_pkg
    fld.ref_sphere = rs_pkg
    wf = wavefront_from_model(opt_model, fld)
    # trace.trace_grid forces check_apertures (trace.py:583); trace_base's default vignetting
    pk = _trace_pupil(opt_model, fld, wvl, dict(check_apertures=True, apply_vignetting=True),
                      None, None, grid=make_grid((-1., -1.), (1., 1.), num_rays),
                      out_mode=abi.OUT_OPD, wf=wf)
    ok = pk.status == abi.OK
    with np.errstate(invalid='ignore'):
        opd = np.where(ok, pk.seg[0, 0] / opt_model.nm_to_sys_units(wvl), 0.0)
    return np.stack([pk.pupil[0], pk.pupil[1], opd], axis=1).reshape(num_rays, num_rays, 3)


class _DeferredWavefront:
    """what the fused :func:`trace_wavefront` hands to :func:`focus_wavefront`
    in place of a grid of ray packets: the grid definition and trace options.
    The trace does not depend on the focus, so re-evaluating trace + OPD on the
    device for every refocus gives what the reference's pre-calc / refocus split
    gives (same formulas, same operation order, waveabr.py:309-353)."""

    def __init__(self, grid_def, kwargs, ctx=None):
        self.grid_def = grid_def
        self.kwargs = kwargs
        self._ctx = ctx             # (opt_model, fld, wvl, foc) for consumers of the grid itself
        self._grid = None

    # any other consumer of RayGrid.grid_pkg[0] sees the reference's grid of
    # [px, py, ray_pkg] rows, traced on first use (FULL packets)
    def _materialise(self):
        if self._grid is None:
            opt_model, fld, wvl, foc = self._ctx
            self._grid = trace_ray_grid(opt_model, self.grid_def, fld, wvl, foc,
                                        **dict(self.kwargs))
        return self._grid

    def __iter__(self):
        return iter(self._materialise())

    def __len__(self):
        return len(self._materialise())

    def __getitem__(self, i):
        return self._materialise()[i]


class _DeferredPreCalc:
    """the second half of the reference's trace_wavefront result (the
    wave_abr_pre_calc grid, rayoptics/raytr/analyses.py:755-764), computed from the
    materialised packets only if a consumer other than focus_wavefront asks"""

    def __init__(self, deferred, cr_pkg, ref_sphere):
        self._d, self._cr, self._rs, self._upd = deferred, cr_pkg, ref_sphere, None

    def _materialise(self):
        if self._upd is None:
            from rayoptics.raytr import waveabr
            opt_model, fld, wvl, foc = self._d._ctx
            fod = opt_model['analysis_results']['parax_data'].fod
            self._upd = [[waveabr.wave_abr_pre_calc(fod, fld, wvl, foc, pkg, self._cr, self._rs)
                          if pkg is not None else None for _px, _py, pkg in row]
                         for row in self._d._materialise()]
        return self._upd

    def __iter__(self):
        return iter(self._materialise())

    def __len__(self):
        return len(self._materialise())

    def __getitem__(self, i):
        return self._materialise()[i]


def _opd_fusable(kwargs):
    return (kwargs.get('output_filter') is None and kwargs.get('rayerr_filter') is None
            and not kwargs.get('filter_out_phantoms', False))


def _opd_grid(opt_model, fld, wvl, grid_def, kwargs, wf, value_if_none):
    kw = dict(kwargs)
    for k in ('output_filter', 'rayerr_filter', 'oversize'):
        kw.pop(k, None)
    kw['apply_vignetting'] = kw.get('apply_vignetting', False)      # trace_ray_grid :674
    num = grid_def[2]
    pk = _trace_pupil(opt_model, fld, wvl, kw, None, None,
                      grid=make_grid(grid_def[0], grid_def[1], num),
                      out_mode=abi.OUT_OPD, wf=wf)
    convert_to_opd = 1 / opt_model.nm_to_sys_units(wvl)
    ok = pk.status == abi.OK
    opd = np.where(ok, convert_to_opd * pk.seg[0, 0], value_if_none)
    return np.stack([pk.pupil[0], pk.pupil[1], opd], axis=1).reshape(num, num, 3)


def trace_wavefront(opt_model, fld, wvl, foc, image_pt_2d=None, image_delta=None,
                    num_rays=21, **kwargs):
    """rayoptics/raytr/analyses.py:735-766"""
    from .table import wavefront_from_model, UnsupportedModelError
    ref_sphere, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, foc, image_pt_2d, image_delta)
    fld.chief_ray = cr_pkg
    fld.ref_sphere = ref_sphere
    oversize = kwargs.get('oversize', 1.)
    vig_bbox = fld.vignetting_bbox(opt_model['osp']['pupil'], oversize=oversize)
    grid_def = [vig_bbox[0], vig_bbox[1], num_rays]
    kwargs['check_apertures'] = kwargs.get('check_apertures', True)
    if _opd_fusable(kwargs):
        try:
            wavefront_from_model(opt_model, fld)         # finite reference sphere?
            d = _DeferredWavefront(grid_def, dict(kwargs), (opt_model, fld, wvl, foc))
            return d, _DeferredPreCalc(d, cr_pkg, ref_sphere)
        except UnsupportedModelError:
            pass
    from rayoptics.raytr import waveabr
    fod = opt_model['analysis_results']['parax_data'].fod
    grid = trace_ray_grid(opt_model, grid_def, fld, wvl, foc, **kwargs)
    upd_grid = [[waveabr.wave_abr_pre_calc(fod, fld, wvl, foc, pkg, cr_pkg, ref_sphere)
                 if pkg is not None else None for _px, _py, pkg in row] for row in grid]
    return grid, upd_grid


def focus_wavefront(opt_model, grid_pkg, fld, wvl, foc, image_pt_2d=None,
                    image_delta=None, value_if_none=np.nan, **kwargs):
    """rayoptics/raytr/analyses.py:769-791"""
    from .table import wavefront_from_model, UnsupportedModelError
    grid, upd_grid = grid_pkg
    ref_sphere, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, foc, image_pt_2d, image_delta)
    if isinstance(grid, _DeferredWavefront):
        try:
            own = getattr(fld, 'rox_wavefront', None)   # table-backed models: prebuilt
            wf = own if own is not None else wavefront_from_model(opt_model, fld, cr_pkg, ref_sphere)
            if wf.kind == abi.WF_INF_FULL:
                # RayGrid's route is wave_abr_pre_calc + wave_abr_calc (waveabr.py:427-488):
                # on an infinite reference sphere the final sum is associated differently
                # from wave_abr_full_calc_inf_ref's
                wf = abi.Wavefront.from_buffer_copy(bytes(wf))
                wf.kind = abi.WF_INF_SPLIT
            return _opd_grid(opt_model, fld, wvl, grid.grid_def, grid.kwargs, wf, value_if_none)
        except UnsupportedModelError:       # the sphere went infinite at this focus
            from rayoptics.raytr import waveabr
            fod = opt_model['analysis_results']['parax_data'].fod
            g = trace_ray_grid(opt_model, grid.grid_def, fld, wvl, foc, **dict(grid.kwargs))
            return np.array([[(px, py, waveabr.wave_abr_full_calc(fod, fld, wvl, foc, pkg, cr_pkg,
                                                                  ref_sphere)
                               / opt_model.nm_to_sys_units(wvl)) if pkg is not None
                              else (px, py, value_if_none) for px, py, pkg in row] for row in g])
    from rayoptics.raytr import waveabr
    fod = opt_model['analysis_results']['parax_data'].fod
    convert_to_opd = 1 / opt_model.nm_to_sys_units(wvl)
    return np.array([[(g[0], g[1], convert_to_opd * waveabr.wave_abr_calc(
        fod, fld, wvl, foc, g[2], cr_pkg, u, ref_sphere)) if g[2] is not None
        else (g[0], g[1], value_if_none) for g, u in zip(ig, iu)] for ig, iu in zip(grid, upd_grid)])


class _DeferredRayList:
    """what the fused :func:`trace_pupil_coords` hands to
    :func:`focus_pupil_coords`: the pupil coordinates and trace options.  It
    still behaves as the reference's list of ``[px, py, ray_pkg]`` entries for
    any other consumer (materialised by a FULL device trace on first use)."""

    def __init__(self, opt_model, pupil_coords, fld, wvl, foc, kwargs):
        self.opt_model, self.fld, self.wvl, self.foc = opt_model, fld, wvl, foc
        self.pupil = np.array([[p[0], p[1]] for p in pupil_coords], dtype=float).reshape(-1, 2)
        self.kwargs = kwargs
        self._list = None

    def _materialise(self):
        if self._list is None:
            self._list = trace_ray_list(self.opt_model, [p.copy() for p in self.pupil],
                                        self.fld, self.wvl, self.foc, **dict(self.kwargs))
        return self._list

    def __iter__(self):
        return iter(self._materialise())

    def __len__(self):
        return len(self._materialise())

    def __getitem__(self, i):
        return self._materialise()[i]


def trace_pupil_coords(opt_model, pupil_coords, fld, wvl, foc,
                       image_pt_2d=None, image_delta=None, **kwargs):
    """rayoptics/raytr/analyses.py:545-558"""
    ref_sphere, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, foc, image_pt_2d, image_delta)
    fld.chief_ray = cr_pkg
    fld.ref_sphere = ref_sphere
    kwargs['check_apertures'] = kwargs.get('check_apertures', True)
    if _opd_fusable(kwargs) and not kwargs.get('append_if_none', False):
        return _DeferredRayList(opt_model, pupil_coords, fld, wvl, foc, dict(kwargs))
    return trace_ray_list(opt_model, pupil_coords, fld, wvl, foc, **kwargs)


def focus_pupil_coords(opt_model, ray_list, fld, wvl, foc,
                       image_pt_2d=None, image_delta=None, **kwargs):
    """rayoptics/raytr/analyses.py:561-580: transverse aberrations of
    pre-traced rays at a (new) focus.  For the deferred list this is one HITS
    launch over the pupil coordinates (the trace does not depend on focus)."""
    ref_sphere, _cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, foc, image_pt_2d, image_delta)
    image_pt = ref_sphere[0]
    if isinstance(ray_list, _DeferredRayList):
        kw = dict(ray_list.kwargs)
        for k in ('output_filter', 'rayerr_filter', 'append_if_none'):
            kw.pop(k, None)
        kw['apply_vignetting'] = kw.get('apply_vignetting', True)
        eng, f, wi, opts = _launch_setup(opt_model, fld, wvl, kw, abi.OUT_HITS_COMPACT,
                                         foc, image_pt[:2])
        return eng.trace_pupil_list_hits(f, ray_list.pupil[:, 0].copy(),
                                         ray_list.pupil[:, 1].copy(), wi, opts)
    data = []
    for _px, _py, pkg in ray_list:
        if pkg is not None:
            seg = pkg[0][-1]
            dist = foc / seg[1][2]
            t_abr = (seg[0] + dist * seg[1]) - image_pt
            data.append((t_abr[0], t_abr[1]))
        else:
            data.append(np.nan)
    return np.array(data)


# ---- RayFan ------------------------------------------------------------------
def _fan_def(xy, num_rays):
    fan_start = np.array([0., 0.])
    fan_stop = np.array([0., 0.])
    fan_start[xy] = -1.0
    fan_stop[xy] = 1.0
    return [fan_start, fan_stop, num_rays]


def _fan_data(opt_model, fld, wvl, foc, fan_def, kwargs, wf, image_pt):
    """one ROX_OUT_FAN launch -> the reference's fan_data list: ((px, py), (dx, dy, opd))
    for every ray that gets through (trace_ray_fan drops the others, analyses.py:212-230)"""
    kw = dict(kwargs)
    for k in ('output_filter', 'rayerr_filter'):
        kw.pop(k, None)
    kw['apply_vignetting'] = kw.get('apply_vignetting', True)
    pk = _trace_pupil(opt_model, fld, wvl, kw, None, None,
                      grid=make_grid(fan_def[0], fan_def[1], fan_def[2], abi.GRID_FAN),
                      out_mode=abi.OUT_FAN, foc=foc, image_pt=image_pt[:2], wf=wf)
    convert_to_opd = 1 / opt_model.nm_to_sys_units(wvl)
    seg = pk.seg[0]
    return [((pk.pupil[0, r], pk.pupil[1, r]), (seg[0, r], seg[1, r], convert_to_opd * seg[2, r]))
            for r in range(fan_def[2]) if pk.status[r] == abi.OK]


def eval_fan(opt_model, fld, wvl, foc, xy, image_pt_2d=None, image_delta=None, num_rays=21,
             output_filter=None, rayerr_filter=None, **kwargs):
    """rayoptics/raytr/analyses.py:233-274: dx, dy and OPD across a fan"""
    from .table import wavefront_from_model
    ref_sphere, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, foc, image_pt_2d, image_delta)
    fld.chief_ray = cr_pkg
    fld.ref_sphere = ref_sphere
    fan_def = _fan_def(xy, num_rays)
    if output_filter is None and rayerr_filter is None and not kwargs.get('filter_out_phantoms', False):
        wf = wavefront_from_model(opt_model, fld)
        return _fan_data(opt_model, fld, wvl, foc, fan_def, kwargs, wf, ref_sphere[0])
    from rayoptics.raytr import waveabr
    fod = opt_model['analysis_results']['parax_data'].fod
    fan = trace_ray_fan(opt_model, fan_def, fld, wvl, foc, output_filter=output_filter,
                        rayerr_filter=rayerr_filter, **kwargs)
    convert_to_opd = 1 / opt_model.nm_to_sys_units(wvl)
    out = []
    for px, py, pkg in fan:
        if pkg is not None:
            seg = pkg[0][-1]
            t_abr = (seg[0] + (foc / seg[1][2]) * seg[1]) - ref_sphere[0]
            opd = convert_to_opd * waveabr.wave_abr_full_calc(fod, fld, wvl, foc, pkg, cr_pkg, ref_sphere)
            out.append(((px, py), (t_abr[0], t_abr[1], opd)))
        else:
            out.append((px, py, np.nan))
    return out


class _DeferredFan(_DeferredWavefront):
    """what the fused :func:`trace_fan` hands to :func:`focus_fan`: the fan
    definition and trace options (a fan of [px, py, ray_pkg] for anyone else)"""

    def _materialise(self):
        if self._grid is None:
            opt_model, fld, wvl, foc = self._ctx
            kw = dict(self.kwargs)
            self._grid = trace_ray_fan(opt_model, self.grid_def, fld, wvl, foc, **kw)
        return self._grid


class _DeferredFanPreCalc(_DeferredPreCalc):
    def _materialise(self):
        if self._upd is None:
            from rayoptics.raytr import waveabr
            opt_model, fld, wvl, foc = self._d._ctx
            fod = opt_model['analysis_results']['parax_data'].fod
            self._upd = [waveabr.wave_abr_pre_calc(fod, fld, wvl, foc, pkg, self._cr, self._rs)
                         if pkg is not None else None for _px, _py, pkg in self._d._materialise()]
        return self._upd


def trace_fan(opt_model, fld, wvl, foc, xy, image_pt_2d=None, image_delta=None, num_rays=21,
              output_filter=None, rayerr_filter=None, **kwargs):
    """rayoptics/raytr/analyses.py:277-314"""
    ref_sphere, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, foc, image_pt_2d, image_delta)
    fld.chief_ray = cr_pkg
    fld.ref_sphere = ref_sphere
    fan_def = _fan_def(xy, num_rays)
    if output_filter is None and rayerr_filter is None and not kwargs.get('filter_out_phantoms', False):
        d = _DeferredFan(fan_def, dict(kwargs), (opt_model, fld, wvl, foc))
        return d, _DeferredFanPreCalc(d, cr_pkg, ref_sphere)
    from rayoptics.raytr import waveabr
    from rayoptics.raytr import traceerror as terr
    fod = opt_model['analysis_results']['parax_data'].fod
    fan = trace_ray_fan(opt_model, fan_def, fld, wvl, foc, output_filter=output_filter,
                        rayerr_filter=rayerr_filter, **kwargs)
    upd_fan = [waveabr.wave_abr_pre_calc(fod, fld, wvl, foc, pkg, cr_pkg, ref_sphere)
               if pkg is not None and not isinstance(pkg, terr.TraceError) else None
               for _px, _py, pkg in fan]
    return fan, upd_fan


def focus_fan(opt_model, fan_pkg, fld, wvl, foc, image_pt_2d=None, image_delta=None, **kwargs):
    """rayoptics/raytr/analyses.py:317-345"""
    from .table import wavefront_from_model
    fan, upd_fan = fan_pkg
    ref_sphere, cr_pkg = _setup_pupil_coords(opt_model, fld, wvl, foc, image_pt_2d, image_delta)
    if isinstance(fan, _DeferredFan):
        own = getattr(fld, 'rox_wavefront', None)
        wf = own if own is not None else wavefront_from_model(opt_model, fld, cr_pkg, ref_sphere)
        if wf.kind == abi.WF_INF_FULL:          # the pre-calc / calc split, as in focus_wavefront
            wf = abi.Wavefront.from_buffer_copy(bytes(wf))
            wf.kind = abi.WF_INF_SPLIT
        return _fan_data(opt_model, fld, wvl, foc, fan.grid_def, fan.kwargs, wf, ref_sphere[0])
    from rayoptics.raytr import waveabr
    from rayoptics.raytr import traceerror as terr
    fod = opt_model['analysis_results']['parax_data'].fod
    convert_to_opd = 1 / opt_model.nm_to_sys_units(wvl)
    out = []
    for (px, py, pkg), fiu in zip(fan, upd_fan):
        if pkg is not None and not isinstance(pkg, terr.TraceError):
            seg = pkg[0][-1]
            t_abr = (seg[0] + (foc / seg[1][2]) * seg[1]) - ref_sphere[0]
            opd = convert_to_opd * waveabr.wave_abr_calc(fod, fld, wvl, foc, pkg, cr_pkg, fiu, ref_sphere)
            out.append(((px, py), (t_abr[0], t_abr[1], opd)))
        else:
            out.append((px, py, np.nan))
    return out


# ---- point spread function ------------------------------------------------------
PSF_BACKEND = None          # None -> engine.calc_psf (the HIP path); tests inject a double


def calc_psf(wavefront, ndim, maxdim):
    """rayoptics/raytr/analyses.py:848-875: the PSF of an OPD grid -- the zero-padded
    pupil function exp(i 2 pi W), its shifted 2-D FFT, |.|^2, normalised.  One call
    into the library (``rox_calc_psf``: a pruned DFT on the fp64 matrix cores) instead
    of a maxdim x maxdim Python loop and a host FFT; any even ``ndim`` that fits,
    any ``maxdim``."""
    fn = PSF_BACKEND
    if fn is None:
        from .engine import calc_psf as fn
    return fn(wavefront, ndim, maxdim)


def update_psf_data(pupil_grid, build='rebuild'):
    """rayoptics/raytr/analyses.py:878-883"""
    pupil_grid.update_data(build=build)
    return calc_psf(pupil_grid.grid[2], pupil_grid.num_rays, pupil_grid.maxdim)

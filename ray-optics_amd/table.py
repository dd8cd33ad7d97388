"""Flat surface table: the read-only model data the trace kernels consume.

The reference hands ``trace_raw`` a per-wavelength list of path tuples
``(Intfc, Gap, Tfrm, Indx, Zdir)`` (rayoptics/seq/sequential.py:149-202,
rayoptics/optical/model_constants.py:12).  :class:`SurfaceTable` flattens that
list into ``rox_surface`` rows (include/roxtrace.h) plus an ``n_table[W][N]``
of refractive indices, by *reading* a live reference ``SequentialModel`` --
nothing is re-derived: transforms come from ``seq_model.lcl_tfrms``, indices
from ``seq_model.rndx``, ``max_nonzero_coef`` from the profile.

Interfaces the kernels do not implement (unknown profile, aperture or phase
element classes; DiffractiveElements with a phase function other than
``radial_phase_fct``) raise :class:`UnsupportedModelError` so that callers keep
such models on the reference's own CPU path.
"""
import ctypes as C
import json
import numpy as np

import numbers

from . import abi


class UnsupportedModelError(Exception):
    """the model uses an interface kind the device kernels do not cover"""


def _profile_row(row, prof):
    kind = type(prof).__name__
    if kind not in abi.PROFILE_NAMES:
        raise UnsupportedModelError(f'profile {kind} is not supported')
    row.profile = abi.PROFILE_NAMES[kind]
    row.cv = float(prof.cv)
    row.flags = 0
    if kind in ('Spherical', 'Conic') and isinstance(prof.cv, numbers.Integral) and prof.cv == 0:
        # A curvature typed as the *integer* 0 (the reference's own double Gauss data,
        # rayoptics/raytr/tests/ag_dblgauss_s.py): `-self.cv*p[0]` in Spherical/Conic.df
        # (profiles.py:360-362, 605-609) is then `0 * x` = +0 for x > 0, where a float 0.0
        # gives `-0.0 * x` = -0.  The row says so in a flag that only the df expression reads;
        # cv stays +0.0, which is what intersect() and sag() make of the integer (a cv of
        # -0.0 would reproduce df but turn `cv*p.dot(p) - 2*p[2]` into -0.0 at p[2] == 0).
        row.flags = abi.SURF_CV_INT_ZERO
    if kind == 'Spherical':
        row.cc, row.ec = 0.0, 1.0
    elif kind == 'RadialPolynomial':
        # stores ec; cc is the derived property (profiles.py:930-937)
        row.ec = float(prof.ec)
        row.cc = float(prof.cc)
    else:
        row.cc = float(prof.cc)
        row.ec = float(prof.ec)         # property: cc + 1.0 (profiles.py:515-517)
    if kind in ('YToroid', 'XToroid'):
        row.cR = float(prof.cR)
    if kind in ('EvenPolynomial', 'RadialPolynomial', 'YToroid', 'XToroid'):
        coefs = [float(c) for c in prof.coefs]
        # profiles.py:827-832 calc_max_nonzero_coef (refreshed by update())
        mnc = getattr(prof, 'max_nonzero_coef', None)
        if mnc is None:
            mnc = 0
            for i, c in enumerate(coefs):
                if c != 0.0:
                    mnc = i + 1
        if mnc > abi.MAX_COEF:
            raise UnsupportedModelError(
                f'{kind} with {mnc} coefficients (max {abi.MAX_COEF})')
        row.ncoef = int(mnc)
        for i in range(mnc):
            row.coefs[i] = coefs[i]


def _aperture_rows(row, ifc):
    cas = getattr(ifc, 'clear_apertures', None) or []
    if len(cas) > abi.MAX_AP:
        raise UnsupportedModelError(f'{len(cas)} clear apertures on one surface')
    row.n_ap = len(cas)
    for k, ca in enumerate(cas):
        a = row.ap[k]
        kind = type(ca).__name__
        a.is_obscuration = 1 if getattr(ca, 'is_obscuration', False) else 0
        a.x_offset = float(getattr(ca, 'x_offset', 0.0))
        a.y_offset = float(getattr(ca, 'y_offset', 0.0))
        if kind == 'Circular':
            a.kind, a.a, a.b = abi.AP_CIRCULAR, float(ca.radius), 0.0
        elif kind == 'Rectangular':
            a.kind = abi.AP_RECTANGULAR
            a.a, a.b = float(ca.x_half_width), float(ca.y_half_width)
        elif kind == 'Elliptical':
            # no point_inside() in the reference (surface.py:472-494): the base
            # class returns None, so every ray is blocked
            a.kind = abi.AP_ALWAYS_BLOCK
            a.a, a.b = float(ca.x_half_width), float(ca.y_half_width)
        else:
            raise UnsupportedModelError(f'aperture {kind} is not supported')


def _vec3(v):
    return [float(v[0]), float(v[1]), float(v[2])]


def _phase_row(row, pe):
    """rox_phase from ``ifc.phase_element`` (rayoptics/oprops/doe.py)"""
    ph = row.ph
    kind = type(pe).__name__
    if kind == 'DiffractionGrating':            # doe.py:57-175 (phase_ludwig)
        ph.kind = abi.PH_GRATING
        ph.order = float(pe.order)
        ph.spacing_nm = float(pe._grating_spacing_nm)
        for i, v in enumerate(_vec3(pe.grating_normal)):
            ph.a[i] = v
    elif kind == 'DiffractiveElement':          # doe.py:225-323
        fct = getattr(pe.phase_fct, '__name__', None)
        if fct != 'radial_phase_fct':
            raise UnsupportedModelError(f'DiffractiveElement phase function {fct!r}')
        coefs = [float(c) for c in pe.coefficients]
        if len(coefs) > abi.MAX_COEF:
            raise UnsupportedModelError(f'DiffractiveElement with {len(coefs)} coefficients')
        ph.kind = abi.PH_DOE_RADIAL
        ph.ncoef = len(coefs)
        for i, c in enumerate(coefs):
            ph.coefs[i] = c
        ph.order = float(pe.order)
        ph.ref_wl = float(pe.ref_wl)
    elif kind == 'HolographicElement':          # doe.py:326-397
        ph.kind = abi.PH_HOLOGRAM
        ph.ref_wl = float(pe.ref_wl)
        ph.flags = (1 if pe.ref_virtual else 0) | (2 if pe.obj_virtual else 0)
        for i, v in enumerate(_vec3(pe.ref_pt)):
            ph.a[i] = v
        for i, v in enumerate(_vec3(pe.obj_pt)):
            ph.b[i] = v
    else:
        raise UnsupportedModelError(f'phase element {kind} is not supported')


def rt_order_of(rt):
    """which dgemv kernel NumPy's ``rt.dot(v)`` reaches: the F-ordered transpose
    view (``r.transpose()``) or a C-ordered array (include/roxtrace.h ROX_RT_*)"""
    flags = getattr(rt, 'flags', None)
    if flags is not None and flags['F_CONTIGUOUS'] and not flags['C_CONTIGUOUS']:
        return abi.RT_F_ORDER
    return abi.RT_C_ORDER


class SurfaceTable:
    """rows[N] of ``rox_surface`` + n_table[W][N] + the wavelengths (nm)."""

    def __init__(self, rows, n_table, wvls, stop_idx=None):
        self.rows = rows
        self.n_table = np.ascontiguousarray(n_table, dtype=np.float64)
        self.wvls = [float(w) for w in wvls]
        self.wvls_arr = np.ascontiguousarray(self.wvls, dtype=np.float64)   # nm, for the C ABI
        self.stop_idx = stop_idx
        # glass names that ingest.Prescription.to_table gave n = 1.5 for want of dispersion data
        self.fallback_glasses = ()
        assert self.n_table.shape == (len(self.wvls), len(rows))

    @property
    def n_ifcs(self):
        return len(self.rows)

    def wvl_index(self, wvl):
        """rayoptics/seq/sequential.py:281-285 index_for_wavelength: the
        wavelength must be an exact member of the spectral region (a miss raises the
        reference's own error -- ``list.index``'s ValueError naming the value as given)"""
        try:
            return self.wvls.index(float(wvl))
        except (ValueError, TypeError):
            raise ValueError(f'{wvl!r} is not in list') from None

    def has_phantoms(self):
        return any(r.mode == abi.PHANTOM for r in self.rows)

    # -- builders ---------------------------------------------------------
    @classmethod
    def from_paths(cls, paths, wvls, stop_idx=None):
        """paths[w] = list of reference path tuples for wavelength wvls[w]."""
        N = len(paths[0])
        rows = (abi.Surface * N)()
        n_table = np.ones((len(wvls), N))
        prev_zdir = 1.0
        for i, seg in enumerate(paths[0]):
            ifc, _gap, tfrm, _n, zdir = seg
            row = rows[i]
            row.mode = abi.MODE_NAMES.get(ifc.interact_mode, abi.DUMMY)
            if type(ifc).__name__ == 'ThinLens':
                # its own planar intersect / constant normal (thinlens.py:128-136)
                row.profile = abi.THINLENS
                row.ec = 1.0
            elif hasattr(ifc, 'profile'):
                _profile_row(row, ifc.profile)
            else:
                raise UnsupportedModelError(
                    f'interface {type(ifc).__name__} has no surface profile')
            # raytrace.py:205 tests hasattr only: a None phase_element would fail
            # in the reference, so it is refused here
            if hasattr(ifc, 'phase_element'):
                if ifc.phase_element is None:
                    raise UnsupportedModelError('phase_element is None')
                _phase_row(row, ifc.phase_element)
            _aperture_rows(row, ifc)
            row.max_aperture = float(ifc.max_aperture)
            if tfrm is None:
                rt, t = np.identity(3), np.zeros(3)
            else:
                rt, t = tfrm
            row.rt_order = rt_order_of(rt)
            for a in range(3):
                for b in range(3):
                    row.rt[3 * a + b] = float(rt[a][b])
                row.t[a] = float(t[a])
            if zdir is None:
                zdir = prev_zdir
            row.z_dir = float(zdir)
            prev_zdir = float(zdir)
        for w, path in enumerate(paths):
            assert len(path) == N
            for i, seg in enumerate(path):
                n = seg[3]
                n_table[w, i] = 1.0 if n is None else float(n)
        return cls(rows, n_table, wvls, stop_idx)

    @classmethod
    def from_seq_model(cls, seq_model, wvls=None):
        """read a live reference ``SequentialModel`` (duck-typed)."""
        if wvls is None:
            wvls = list(seq_model.opt_model['osp']['wvls'].wavelengths)
        paths = [list(seq_model.path(wl=w)) for w in wvls]
        return cls.from_paths(paths, wvls, getattr(seq_model, 'stop_surface', None))

    @classmethod
    def from_prescription(cls, surfaces, wvls=(587.6,), stop_idx=None):
        """standalone builder for centred systems, no reference needed.

        surfaces: one dict per interface (object first, image last) with keys
        ``cv, thi, n`` (n: float or per-wavelength list; the medium *after*
        the interface) and optional ``mode, profile, cc, ec, coefs,
        max_aperture``.  Transforms follow rayoptics/elem/transform.py:143-166
        for undecentered surfaces: Rt = I, t = (0, 0, thi); z_dir flips after
        each mirror as in rayoptics/seq/sequential.py:640-655.
        """
        N = len(surfaces)
        rows = (abi.Surface * N)()
        n_table = np.ones((len(wvls), N))
        zdir = 1.0
        for i, s in enumerate(surfaces):
            row = rows[i]
            mode = s.get('mode', 'dummy' if i in (0, N - 1) else 'transmit')
            row.mode = abi.MODE_NAMES[mode]
            prof = s.get('profile', 'Spherical')
            row.profile = abi.PROFILE_NAMES[prof]
            row.cv = float(s.get('cv', 0.0))
            if prof == 'RadialPolynomial':
                row.ec = float(s.get('ec', 1.0))
                row.cc = row.ec - 1.0
            else:
                row.cc = float(s.get('cc', 0.0))
                row.ec = row.cc + 1.0
            coefs = list(s.get('coefs', []))
            mnc = 0
            for k, c in enumerate(coefs):
                if c != 0.0:
                    mnc = k + 1
            if mnc > abi.MAX_COEF:
                raise UnsupportedModelError('too many coefficients')
            row.ncoef = mnc
            for k in range(mnc):
                row.coefs[k] = float(coefs[k])
            for a in range(3):
                row.rt[4 * a] = 1.0
            row.rt_order = abi.RT_C_ORDER       # np.identity(3)
            row.t[2] = float(s.get('thi', 0.0)) if i < N - 1 else 0.0
            if mode == 'reflect':
                zdir = -zdir
            row.z_dir = zdir
            row.max_aperture = float(s.get('max_aperture', 1.0))
            n = s.get('n', 1.0)
            for w in range(len(wvls)):
                n_table[w, i] = float(n[w]) if isinstance(n, (list, tuple)) else float(n)
        return cls(rows, n_table, wvls, stop_idx)

    # -- (de)serialisation for golden fixtures ------------------------------
    def to_dict(self):
        rows = []
        for r in self.rows:
            rows.append(dict(
                mode=r.mode, profile=r.profile, ncoef=r.ncoef, n_ap=r.n_ap,
                rt_order=r.rt_order, flags=r.flags,
                cv=r.cv, cc=r.cc, ec=r.ec, cR=r.cR, coefs=list(r.coefs),
                rt=list(r.rt), t=list(r.t), z_dir=r.z_dir,
                max_aperture=r.max_aperture,
                ap=[dict(kind=a.kind, is_obscuration=a.is_obscuration,
                         x_offset=a.x_offset, y_offset=a.y_offset, a=a.a, b=a.b)
                    for a in r.ap[:r.n_ap]]))
            if r.ph.kind != abi.PH_NONE:
                ph = r.ph
                rows[-1]['ph'] = dict(kind=ph.kind, ncoef=ph.ncoef, flags=ph.flags,
                                      order=ph.order, ref_wl=ph.ref_wl,
                                      spacing_nm=ph.spacing_nm, a=list(ph.a), b=list(ph.b),
                                      coefs=list(ph.coefs))
        return dict(wvls=self.wvls, stop_idx=self.stop_idx,
                    n_table=self.n_table.tolist(), rows=rows)

    @classmethod
    def from_dict(cls, d):
        N = len(d['rows'])
        rows = (abi.Surface * N)()
        for row, s in zip(rows, d['rows']):
            row.mode, row.profile = s['mode'], s['profile']
            row.ncoef, row.n_ap = s['ncoef'], s['n_ap']
            row.rt_order = s.get('rt_order', abi.RT_F_ORDER)
            row.cv, row.cc, row.ec = s['cv'], s['cc'], s['ec']
            if 'flags' in s:
                row.flags = s['flags']
            elif (s['profile'] in (abi.SPHERICAL, abi.CONIC) and s['cv'] == 0.0
                  and np.signbit(s['cv'])):
                # tables written before the flag existed spelled an integer-zero curvature
                # as the float -0.0 (whose df has the same zero signs)
                row.cv, row.flags = 0.0, abi.SURF_CV_INT_ZERO
            row.cR = s.get('cR', 0.0)
            for k, c in enumerate(s['coefs']):
                row.coefs[k] = c
            for k, v in enumerate(s['rt']):
                row.rt[k] = v
            for k, v in enumerate(s['t']):
                row.t[k] = v
            row.z_dir, row.max_aperture = s['z_dir'], s['max_aperture']
            for k, a in enumerate(s['ap']):
                ap = row.ap[k]
                ap.kind, ap.is_obscuration = a['kind'], a['is_obscuration']
                ap.x_offset, ap.y_offset = a['x_offset'], a['y_offset']
                ap.a, ap.b = a['a'], a['b']
            if 'ph' in s:
                p, ph = s['ph'], row.ph
                ph.kind, ph.ncoef, ph.flags = p['kind'], p['ncoef'], p['flags']
                ph.order, ph.ref_wl, ph.spacing_nm = p['order'], p['ref_wl'], p['spacing_nm']
                for k in range(3):
                    ph.a[k], ph.b[k] = p['a'][k], p['b'][k]
                for k, c in enumerate(p['coefs']):
                    ph.coefs[k] = c
        return cls(rows, np.array(d['n_table'], dtype=np.float64), d['wvls'],
                   d.get('stop_idx'))

    def save(self, path):
        with open(path, 'w') as f:
            json.dump(self.to_dict(), f)

    @classmethod
    def load(cls, path):
        with open(path) as f:
            return cls.from_dict(json.load(f))


def field_struct(pt0, aim, eprad, z_enp, vig=(0., 0., 0., 0.), z_dir0=1.0,
                 kind=abi.FLD_EPD, rot=None, cr_dir=(0., 0.)):
    """fill a ``rox_field`` (vig = (vlx, vux, vly, vuy)); see include/roxtrace.h
    for what ``eprad`` / ``z_enp`` carry in each kind."""
    f = abi.Field()
    for i in range(3):
        f.pt0[i] = float(pt0[i])
    f.aim[0], f.aim[1] = float(aim[0]), float(aim[1])
    f.eprad, f.z_enp = float(eprad), float(z_enp)
    f.vlx, f.vux, f.vly, f.vuy = (float(v) for v in vig)
    f.z_dir0 = float(z_dir0)
    f.kind = int(kind)
    if rot is not None:
        f.rot_order = rt_order_of(rot)
        for a in range(3):
            for b in range(3):
                f.rot[3 * a + b] = float(rot[a][b])
    f.cr_dir[0], f.cr_dir[1] = float(cr_dir[0]), float(cr_dir[1])
    return f


def _obj_coords(opt_model, fld, cache):
    """``osp.obj_coords(fld)`` (rayoptics/raytr/opticalspec.py:990-1091).  For fields given
    as real image heights it runs ``wideangle.eval_real_image_ht`` -- a reverse chief-ray
    iteration of several single rays -- on every call; the reference calls it once per
    *ray* (trace_base -> ray_start_from_osp), the drop-ins once per launch, and a spot
    diagram's launches for one field all get the same answer.  ``cache`` (a dict that lives
    and dies with the engine handle, i.e. with the model state) memoises exactly that case,
    keyed by everything the result depends on beside the model: the field's coordinates, the
    field specification and the first-order data object; the side effect (``fld.aim_info``
    set to the pupil aim the iteration found, :1019-1030) is replayed on a hit."""
    osp = opt_model['optical_spec']
    fov = osp['fov']
    if cache is None or tuple(fov.key) != ('image', 'real height'):
        return osp.obj_coords(fld)
    # (the entry keeps the field and the first-order data object alive and compares them by
    # identity: an id() alone could be reused by another object after an update_model();
    # the stop surface is part of what eval_real_image_ht reads, wideangle.py:633-634)
    parax = opt_model['analysis_results']['parax_data']
    key = (id(fld), float(fld.x), float(fld.y), float(fov.value), bool(fov.is_relative),
           bool(fov.is_wide_angle), float(osp['wvls'].central_wvl), id(parax),
           opt_model['seq_model'].stop_surface)
    hit = cache.get(key)
    if hit is not None and (hit[3] is not fld or hit[4] is not parax):
        hit = None
    if hit is None:
        p0, d0 = osp.obj_coords(fld)
        if len(cache) >= 64:        # every update_model() brings a new first-order data object:
            cache.clear()           # entries of past updates can never hit again
        hit = cache[key] = (np.array(p0, dtype=float), np.array(d0, dtype=float),
                            None if fld.aim_info is None else np.array(fld.aim_info, dtype=float),
                            fld, parax)
    p0, d0, aim = hit[:3]
    fld.aim_info = None if aim is None else (float(aim) if aim.ndim == 0 else aim.copy())
    return p0.copy(), d0.copy()


def field_from_model(opt_model, fld, pupil_type='rel pupil', cache=None):
    """per-field constants of ``OpticalSpecs.ray_start_from_osp``
    (rayoptics/raytr/opticalspec.py:289-400), every branch: 'epd' pupils
    (plain, wide-angle, 'aim pt'), angular pupils ('NA', 'f/#', 'aim dir').
    The per-ray part of each branch runs on the device (``rox_field.kind``).
    Fields that carry prebuilt constants (``fld.rox_field``: table-backed
    models, :mod:`~.workloads`) use them as is.

    Raises, for a field this layer cannot express: :class:`UnsupportedModelError` (a pupil
    specification without a device branch), the reference's ``TraceError`` classes (the
    reverse chief-ray iteration of a real-image-height field, ``osp.obj_coords``), and what
    the reference's own arithmetic raises on degenerate specifications (``TypeError`` for an
    angular pupil with neither chief-ray direction nor aim point, ``ValueError``,
    ``ZeroDivisionError``, ``FloatingPointError``).  The drop-ins answer such a field through the
    reference's own function (``trace._FIELD_ERRORS``)."""
    pre = getattr(fld, 'rox_field', None)
    if pre is not None and pupil_type == 'rel pupil':
        return pre
    # inside one drop-in call (session.hold) the same field is asked for several times: the
    # constants depend, beyond the held model state, on what a chief-ray request may change
    # in between -- the aim and the vignetting factors
    from . import session
    memo = session.held_memo(opt_model)
    if memo is not None:
        ai = getattr(fld, 'aim_info', None)
        mkey = ('field', id(fld), pupil_type, None if ai is None else tuple(np.ravel(ai).tolist()),
                fld.vlx, fld.vux, fld.vly, fld.vuy)
        hit = memo.get(mkey)
        if hit is None:
            hit = memo[mkey] = _field_from_model(opt_model, fld, pupil_type, cache)
        return hit
    return _field_from_model(opt_model, fld, pupil_type, cache)


def _field_from_model(opt_model, fld, pupil_type, cache):
    osp = opt_model['optical_spec']
    fod = opt_model['analysis_results']['parax_data'].fod
    pupil_oi_key, pupil_value_key = osp['pupil'].key
    pupil_value = osp['pupil'].value
    n_obj, n_img = osp.obj_img_rindex()
    p0, d0 = _obj_coords(opt_model, fld, cache)       # :306
    if pupil_oi_key == 'image':                        # :311-325
        if abs(fod.m) < 1e-10:
            pupil_value_key, pupil_value = 'epd', 2 * fod.enp_radius
        elif abs(fod.enp_dist) > 1e10:                 # telecentric entrance pupil
            from rayoptics.parax import etendue
            pupil_value_key = 'NA'
            slp0 = etendue.na2slp_parax(fod.obj_na, n=n_obj)
            pupil_value = etendue.slp2na(slp0, n=n_obj)
        else:
            pupil_value_key, pupil_value = 'epd', 2 * fod.enp_radius
    aim_info = getattr(fld, 'aim_info', None)
    z_enp = fod.enp_dist
    vig = (fld.vlx, fld.vux, fld.vly, fld.vuy)
    z_dir0 = opt_model['seq_model'].z_dir[0]
    if osp['fov'].is_wide_angle:
        # trace.py:302-308: a wide-angle model never flips dir0 and never intersects the
        # object surface, whatever the pupil specification -- z_dir0 = 0 encodes "never flip"
        # (include/roxtrace.h, rox_field.z_dir0)
        z_dir0 = 0.0
    if pupil_value_key == 'epd':
        if pupil_type == 'aim pt':                     # :334-337
            return field_struct(p0, (0., 0.), 0.0, fod.obj_dist + z_enp, vig, z_dir0,
                                kind=abi.FLD_AIM_PT)
        eprad = pupil_value / 2
        if osp['fov'].is_wide_angle:                   # :340-356
            from rayoptics.util.misc_math import rot_v1_into_v2
            rot_mat_d2s = rot_v1_into_v2(d0, np.array([0., 0., 1.]))
            if aim_info is not None:
                z_enp = aim_info
            obj2enp_dist = -(fod.obj_dist + z_enp)
            if osp.conjugate_type('object') == 'infinite':
                enp_pt = np.array([0., 0., obj2enp_dist])
                rot_mat_s2d = rot_v1_into_v2(np.array([0., 0., 1.]), d0)
                pt0 = np.matmul(rot_mat_s2d, enp_pt) - enp_pt
            else:
                pt0 = p0
            return field_struct(pt0, (0., 0.), eprad, obj2enp_dist, vig, z_dir0,
                                kind=abi.FLD_EPD_WIDE, rot=rot_mat_d2s)
        aim_pt = [0., 0.] if aim_info is None else aim_info          # :358-366
        obj2enp_dist = -(fod.obj_dist + z_enp)
        pt0 = obj2enp_dist * np.array([d0[0] / d0[2], d0[1] / d0[2], 0.])
        return field_struct(pt0, aim_pt, eprad, fod.obj_dist + z_enp, vig, z_dir0)
    # an angular based measure, :368-398
    if pupil_type == 'aim dir':
        return field_struct(p0, (0., 0.), 0.0, 0.0, vig, z_dir0, kind=abi.FLD_AIM_DIR)
    if 'NA' in pupil_value_key:
        n = n_obj if pupil_oi_key == 'object' else n_img
        kind, scale = abi.FLD_NA, pupil_value / n
    elif 'f/#' in pupil_value_key:
        kind, scale = abi.FLD_FNO, -1 / (2 * pupil_value)
    else:
        raise UnsupportedModelError(f'pupil key {pupil_value_key!r}')
    if d0 is not None:
        cr_dir = d0[:2]
    else:
        from rayoptics.util.misc_math import normalize
        pt1 = np.array([aim_info[0], aim_info[1], fod.obj_dist + fod.enp_dist])
        cr_dir = normalize(pt1 - p0)[:2]
    return field_struct(p0, (0., 0.), scale, 0.0, vig, z_dir0, kind=kind, cr_dir=cr_dir)


def wavefront_from_model(opt_model, fld, chief_ray_pkg=None, ref_sphere=None):
    """``rox_wavefront``: the chief-ray package and reference sphere that
    ``trace.setup_pupil_coords`` leaves in ``fld.chief_ray`` / ``fld.ref_sphere``
    (rayoptics/raytr/trace.py:608-624), as ``wave_abr_full_calc_finite_pup``
    (rayoptics/raytr/waveabr.py:256-307) or, on an infinite reference sphere
    (``is_kinda_big``, waveabr.py:213-216), ``wave_abr_full_calc_inf_ref``
    (:356-424) reads them."""
    pre = getattr(fld, 'rox_wavefront', None)
    if pre is not None and chief_ray_pkg is None and ref_sphere is None:
        return pre                                  # table-backed models (workloads.TableField)
    fod = opt_model['analysis_results']['parax_data'].fod
    cr_pkg = fld.chief_ray if chief_ray_pkg is None else chief_ray_pkg
    rs = fld.ref_sphere if ref_sphere is None else ref_sphere
    cr, cr_exp_seg = cr_pkg
    cr_ray, cr_op, _wvl = cr
    cr_exp_pt, _cr_exp_dir, cr_exp_dist, ifc, _b4_pt, _b4_dir = cr_exp_seg
    image_pt, ref_dir, ref_radius, lcl_tfrm_last = rs
    w = abi.Wavefront()
    if np.isinf(ref_radius) or abs(ref_radius) > 1e8:       # is_kinda_big, misc_math.py:22-29
        # wave_abr_full_calc_inf_ref (waveabr.py:356-424): the chief-ray-only terms
        # are formed here exactly as the reference forms them per ray
        w.kind = abi.WF_INF_FULL
        if lcl_tfrm_last is not None:
            rt, t = lcl_tfrm_last
            w.last_kind = 1
            w.last_order = rt_order_of(rt)
            for a in range(3):
                for b in range(3):
                    w.last_rt[3 * a + b] = float(rt[a][b])
                w.last_t[a] = float(t[a])
            p_cr_b4, d_cr_b4 = rt.dot(cr_ray[-2][0] - t), rt.dot(cr_ray[-2][1])
        else:
            p_cr_b4, d_cr_b4 = cr_ray[-2][0], cr_ray[-2][1]
        op_cr_b4 = np.dot(d_cr_b4, -p_cr_b4)                # ray_dist_to_perp_from_origin
        w.v_be = float(cr_op + op_cr_b4)
        for i in range(3):
            w.d_cr_b4[i] = float(d_cr_b4[i])
            w.cr_last_p[i] = float(cr_ray[-1][0][i])
            w.cr_last_d[i] = float(cr_ray[-1][1][i])
            w.image_pt[i] = float(image_pt[i])
    for i in range(3):
        w.cr1_p[i] = float(cr_ray[1][0][i])
        w.cr0_d[i] = float(cr_ray[0][1][i])
        w.crk_p[i] = float(cr_ray[-2][0][i])
        w.crk_d[i] = float(cr_ray[-2][1][i])
        w.cr_exp_pt[i] = float(cr_exp_pt[i])
        w.ref_dir[i] = float(ref_dir[i])
    w.cr_op = float(cr_op)
    w.cr_exp_dist = float(cr_exp_dist)
    w.ref_radius = float(ref_radius)
    w.n_obj, w.n_img = abs(float(fod.n_obj)), abs(float(fod.n_img))
    w.sign_soln = -1.0 if ref_dir[2] * cr_ray[-1][1][2] < 0 else 1.0
    # transform_after_surface(ifc, .), rayoptics/elem/transform.py:234-258
    w.after_kind = 0
    dec = getattr(ifc, 'decenter', None)
    if dec:
        r, t = dec.tform_after_surf()
        for i in range(3):
            w.after_t[i] = float(t[i])
        if r is None:
            w.after_kind = 1
        else:
            w.after_kind = 2
            rt = r.transpose()
            w.after_order = rt_order_of(rt)
            for a in range(3):
                for b in range(3):
                    w.after_rt[3 * a + b] = float(rt[a][b])
    return w


def wavefront_to_array(w):
    return np.frombuffer(bytes(w), dtype=np.uint8).copy()


def wavefront_from_array(a):
    b = np.asarray(a, dtype=np.uint8).tobytes()
    n = C.sizeof(abi.Wavefront)
    return abi.Wavefront.from_buffer_copy(b + bytes(max(n - len(b), 0)))   # (ABI v2 fixtures: finite)

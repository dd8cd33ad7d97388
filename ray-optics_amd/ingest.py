"""Prescription files -> :class:`~.table.SurfaceTable`, without the reference's
object model: the on-disk formats either side of
the trace path, read with the standard library only.

  read_zmx   Zemax .zmx    <- rayoptics/zemax/zmxread.py:93-456, 540-600
  read_seq   CODE V .seq   <- rayoptics/codev/cmdproc.py:56-99, 371-450, 544-580
  read_roa   ray-optics .roa (JSON) <- rayoptics/gui/roafile.py:106-123 (structure:
             SURVEY.md section 8c)

Each reader returns a :class:`Prescription` (interfaces, gaps, stop, wavelengths);
``Prescription.to_table()`` flattens it exactly as ``SurfaceTable.from_seq_model``
flattens the live model the reference's importer would have built: same
profile parameters, apertures, interact modes, z_dir bookkeeping
(rayoptics/seq/sequential.py:611-668) and local transforms
(rayoptics/elem/transform.py:86-118, 143-166 for undecentered interfaces).

Scope: STANDARD / EVENASPH / CONI / MIRROR / PARAXIAL / DGRATING / COORDBRK /
TOROIDAL surfaces of .zmx; S / SO / SI / STO / SPH / CON / ASP / K / A..J / RDY / CUY / THI /
REFL / WL, the aperture commands CIR / REX / REY / ELX / ELY / ADX / ADY and the
decenter commands XDE / YDE / ZDE / ADE / BDE / CDE / DAR / BEN / REV of .seq;
Spherical / Conic / EvenPolynomial / RadialPolynomial profiles, Circular /
Rectangular apertures and ``decenter`` records of .roa.

Decentered and tilted interfaces: the local transforms are composed exactly as
``compute_local_transforms`` / ``forward_transform`` compose them
(rayoptics/elem/transform.py:79-166) from ``DecenterData``'s before/after
transforms (rayoptics/elem/surface.py:312-337), with the same NumPy calls so the
rounding is the reference's.  The tilt matrix itself is
``transforms3d.euler.euler2mat(..., 'rxyz')`` in the reference (a third-party
package, absent from the build container): restated as Rx.Ry.Rz -- tilt parity
is unpinned.

Refractive indices: the engine consumes evaluated indices, so a glass *name* has
to be turned into n(wavelength) here.  ``index_of(name, wvl_nm)`` is the hook.
The default, :func:`nominal_index`, evaluates the dispersion formulas on file
(:data:`SELLMEIER`, checked against the CODE V listing
rayoptics/codev/tests/ag_dblgauss.lis:30-34 to 1e-5 and against the catalogue
nd / vd of each glass) and the index samples a file defines itself; a glass
it does not know gets the reference's own not-in-catalogue value n = 1.5
(rayoptics/seq/medium.py:172-203) **with an** :class:`UnknownGlassWarning`, and the
table records it in ``SurfaceTable.fallback_glasses``.
:func:`reference_fallback_index` (pass it explicitly) gives *every* named glass
n = 1.5: that is what the reference's importers produce when ``opticalglass``
has an empty catalogue -- the configuration of the build container, and the one
the importer-parity tests pin.  In a real installation ``opticalglass`` resolves
catalogue glasses to dispersive indices: hand such a lookup in as ``index_of``.
"""
import json
import math
import os
import re

import numpy as np

from . import abi
from .table import SurfaceTable, UnsupportedModelError, rt_order_of

# ---------------------------------------------------------------- dispersion
# Sellmeier-1 coefficients (B1 B2 B3 C1 C2 C3, wavelength in micrometres): nominal
# catalogue values of the glasses named by the reference's fixture files and of the
# most common Schott types.  ND_VD holds the catalogue (nd, vd) of each entry;
# tests/test_ingest_glasses.py checks every formula against it (1e-5 / 0.03).
SELLMEIER = {
    'N-BK7': (1.03961212, 0.231792344, 1.01046945, 0.00600069867, 0.0200179144, 103.560653),
    'N-SK2': (1.28189012, 0.257738258, 0.96818604, 0.0072719164, 0.0242823527, 110.377773),
    'N-SK16': (1.34317774, 0.241144399, 0.994317969, 0.00704687339, 0.0229005, 92.7508526),
    'N-SSK2': (1.4306027, 0.153150554, 1.01390904, 0.00823982975, 0.0333736841, 106.870822),
    'F5': (1.3104463, 0.19603426, 0.96612977, 0.00958633048, 0.0457627627, 115.011883),
    # fused silica, Malitson 1965
    'SILICA': (0.6961663, 0.4079426, 0.8974794, 0.0684043 ** 2, 0.1162414 ** 2, 9.896161 ** 2),
    'N-BAK1': (1.12365662, 0.309276848, 0.881511957, 0.00644742752, 0.0222284402, 107.297751),
    'N-BAK4': (1.28834642, 0.132817724, 0.945395373, 0.00779980626, 0.0315631177, 105.965875),
    'N-BALF4': (1.31004128, 0.142038259, 0.964929351, 0.0079659645, 0.0330672072, 109.19732),
    'N-BAF10': (1.5851495, 0.143559385, 1.08521269, 0.00926681282, 0.0424489805, 105.613573),
    'N-K5': (1.08511833, 0.199562005, 0.930511663, 0.00661099503, 0.024110866, 111.982777),
    'N-FK5': (0.844309338, 0.344147824, 0.910790213, 0.00475111955, 0.0149814849, 97.8600293),
    'N-PSK53A': (1.38121836, 0.196745645, 0.886089205, 0.00706416337, 0.0233251345, 97.4847345),
    'N-KZFS4': (1.35055424, 0.197575506, 1.09962992, 0.0087628207, 0.0371767201, 90.3866994),
    'N-SK4': (1.32993741, 0.228542996, 0.988465211, 0.00716874107, 0.0246455892, 100.886364),
    'N-SK5': (0.991463823, 0.495982121, 0.987393925, 0.00522730467, 0.0172733646, 98.3594579),
    'N-SK14': (0.936155374, 0.594052018, 1.04374583, 0.00461716525, 0.016885927, 103.736265),
    'N-LAK9': (1.46231905, 0.344399589, 1.15508372, 0.00724270156, 0.0243353131, 85.4686868),
    'N-LAK22': (1.14229781, 0.535138441, 1.04088385, 0.00585778594, 0.0198546147, 100.834017),
    'N-LAF2': (1.80984227, 0.15729555, 1.0930037, 0.0101711622, 0.0442431765, 100.687748),
    'N-LASF9': (2.00029547, 0.298926886, 1.80691843, 0.0121426017, 0.0538736236, 156.530829),
    'F2': (1.34533359, 0.209073176, 0.937357162, 0.00997743871, 0.0470450767, 111.886764),
    'N-F2': (1.39757037, 0.159201403, 1.2686543, 0.00995906143, 0.0546931752, 119.248346),
    'N-SF1': (1.60865158, 0.237725916, 1.51530653, 0.0119654879, 0.0590589722, 135.521676),
    'N-SF2': (1.47343127, 0.163681849, 1.36920899, 0.0109019098, 0.0585683687, 127.404933),
    'N-SF4': (1.67780282, 0.282849893, 1.63539276, 0.012679345, 0.0602038419, 145.760496),
    'N-SF5': (1.52481889, 0.187085527, 1.42729015, 0.011254756, 0.0588995392, 129.141675),
    'N-SF6': (1.77931763, 0.338149866, 2.08734474, 0.0133714182, 0.0617533621, 174.01759),
    'SF6': (1.72448482, 0.390104889, 1.04572858, 0.0134871947, 0.0569318095, 118.557185),
    'N-SF8': (1.55075812, 0.209816918, 1.46205491, 0.0114338344, 0.0582725652, 133.24165),
    'N-SF10': (1.62153902, 0.256287842, 1.64447552, 0.0122241457, 0.0595736775, 147.468793),
    'N-SF11': (1.73759695, 0.313747346, 1.89878101, 0.013188707, 0.0623068142, 155.23629),
    'N-SF57': (1.87543831, 0.37375749, 2.30001797, 0.0141749518, 0.0640509927, 177.389795),
}
ND_VD = {
    'N-BK7': (1.51680, 64.17), 'N-SK2': (1.60738, 56.65), 'N-SK16': (1.62041, 60.32),
    'N-SSK2': (1.62229, 53.27), 'F5': (1.60342, 38.03), 'SILICA': (1.45846, 67.82),
    'N-BAK1': (1.57250, 57.55), 'N-BAK4': (1.56883, 55.98), 'N-BALF4': (1.57956, 53.87),
    'N-BAF10': (1.67003, 47.11), 'N-K5': (1.52249, 59.48), 'N-FK5': (1.48749, 70.41),
    'N-PSK53A': (1.61800, 63.39), 'N-KZFS4': (1.61336, 44.49), 'N-SK4': (1.61272, 58.63),
    'N-SK5': (1.58913, 61.27), 'N-SK14': (1.60311, 60.60), 'N-LAK9': (1.69100, 54.71),
    'N-LAK22': (1.65113, 55.89), 'N-LAF2': (1.74397, 44.85), 'N-LASF9': (1.85025, 32.17),
    'F2': (1.62004, 36.37), 'N-F2': (1.62005, 36.43), 'N-SF1': (1.71736, 29.62),
    'N-SF2': (1.64769, 33.82), 'N-SF4': (1.75513, 27.38), 'N-SF5': (1.67271, 32.25),
    'N-SF6': (1.80518, 25.36), 'SF6': (1.80518, 25.43), 'N-SF8': (1.68894, 31.31),
    'N-SF10': (1.72828, 28.53), 'N-SF11': (1.78472, 25.68), 'N-SF57': (1.84666, 23.78),
}
# (BK7 -> N-BK7: Schott lists the lead-free melt with the classic BK7 dispersion constants,
# nd 1.51680 / vd 64.17 for both)
# (BK7 -> N-BK7: the lead-free successor melt; SCHOTT's data sheets give both the same Sellmeier
# coefficients, nd 1.51680, vd 64.17 -- the only dispersion data this module carries for either)
ALIASES = {'BK7': 'N-BK7', 'FUSEDSILICA': 'SILICA', 'F_SILICA': 'SILICA'}


def _canon(name):
    """NSK16_SCHOTT / N-SK16 / nsk16 -> N-SK16.  Aliases are looked up on the whole name first
    (F_SILICA contains the separator the catalogue suffix is split at), then on the name
    without its catalogue suffix (BK7_SCHOTT)"""
    full = name.upper().strip()
    if full in ALIASES:
        return ALIASES[full]
    n = full.split('_')[0]
    n = ALIASES.get(n, n)
    if n in SELLMEIER:
        return n
    if n.startswith('N') and not n.startswith('N-') and ('N-' + n[1:]) in SELLMEIER:
        return 'N-' + n[1:]
    return n


def sellmeier_index(name, wvl_nm):
    b1, b2, b3, c1, c2, c3 = SELLMEIER[_canon(name)]
    l2 = (wvl_nm * 1e-3) ** 2
    return math.sqrt(1.0 + b1 * l2 / (l2 - c1) + b2 * l2 / (l2 - c2) + b3 * l2 / (l2 - c3))


def model_glass_index(nd, vd, wvl_nm):
    """a glass given by (nd, vd) only (Zemax model glasses, CODE V fictitious
    ``nnn.vvv`` codes): two-term Cauchy n = A + B / lambda^2 through nd with
    nF - nC = (nd - 1) / vd.  ``opticalglass.modelglass.ModelGlass`` (absent here)
    fits a Buchdahl model instead -- catalogue parity is unpinned."""
    if not vd:
        return float(nd)
    inv2 = lambda w: 1.0 / (w * 1e-3) ** 2      # noqa: E731
    b = (nd - 1.0) / vd / (inv2(486.1327) - inv2(656.2725))
    return float(nd + b * (inv2(wvl_nm) - inv2(587.5618)))


class UnknownGlassWarning(UserWarning):
    """a named glass without dispersion data on file was given n = 1.5"""


def reference_fallback_index(name, wvl_nm):
    """what the reference's importers give a glass its catalogue does not have"""
    return 1.5


def knows_glass(name):
    return _canon(name) in SELLMEIER


def nominal_index(name, wvl_nm):
    """dispersion formula where one is on file, the reference's fallback otherwise
    (``Prescription.to_table`` warns about and records the glasses that fell back)"""
    if _canon(name) in SELLMEIER:
        return sellmeier_index(name, wvl_nm)
    return 1.5


# ---------------------------------------------------------------- data model
class Ifc:
    """one interface as the importers leave it"""

    def __init__(self):
        self.mode = 'transmit'
        self.profile = 'Spherical'
        self.cv = 0.0
        self.cc = 0.0
        self.coefs = []
        self.max_aperture = 1.0
        self.apertures = []         # dicts: kind, radius | x_half_width, y_half_width, offsets, obsc.
        self.phase = None           # dict for rox_phase
        self.thinlens_power = None
        self.z_type = 'STANDARD'
        self.decenter = None        # dict(dtype, dec[3], euler[3]) <- surface.py DecenterData


def new_decenter(dtype='decenter'):
    return dict(dtype=dtype, dec=[0., 0., 0.], euler=[0., 0., 0.])


def _euler2rot3d(euler):
    """rayoptics/util/misc_math.py:151-161: euler2mat(*deg2rad(-alpha, -beta, gamma), 'rxyz')
    = Rx(ai).Ry(aj).Rz(ak) (transforms3d; restated, see the module docstring)"""
    ai, aj, ak = np.deg2rad(np.array([-euler[0], -euler[1], euler[2]]))
    cx, sx = math.cos(ai), math.sin(ai)
    cy, sy = math.cos(aj), math.sin(aj)
    cz, sz = math.cos(ak), math.sin(ak)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=float)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=float)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=float)
    return rx @ ry @ rz


def _dec_arrays(d):
    """(rot_mat or None, dec) of a decenter record: DecenterData.update, surface.py:312-316"""
    euler = np.array([float(v) for v in d['euler']])
    rot = _euler2rot3d(euler) if euler.any() else None
    return rot, np.array([float(v) for v in d['dec']])


def _tform_before_surf(d):          # surface.py:322-326
    rot, dec = _dec_arrays(d)
    if d['dtype'] != 'reverse':
        return rot, dec
    return None, np.array([0., 0., 0.])


def _tform_after_surf(d):           # surface.py:328-337
    rot, dec = _dec_arrays(d)
    if d['dtype'] in ('reverse', 'dec and return'):
        return (rot.transpose() if rot is not None else None), -dec
    if d['dtype'] == 'bend':
        return rot, np.array([0., 0., 0.])
    return None, np.array([0., 0., 0.])


def forward_transform(dec1, zdist, dec2):
    """rayoptics/elem/transform.py:143-166: rotation and translation from the frame of
    interface 1 (decenter record dec1 or None) to that of interface 2"""
    t_orig = np.array([0., 0., zdist])
    r_after_s1 = r_before_s2 = None
    if dec1:
        r_after_s1, t_after_s1 = _tform_after_surf(dec1)
        t_orig += t_after_s1
    if dec2:
        r_before_s2, t_before_s2 = _tform_before_surf(dec2)
        t_orig += t_before_s2
    r_cascade = np.identity(3)
    if r_after_s1 is not None:
        t_orig = np.matmul(r_after_s1, t_orig)
        r_cascade = r_after_s1
        if r_before_s2 is not None:
            r_cascade = np.matmul(r_after_s1, r_before_s2)
    elif r_before_s2 is not None:
        r_cascade = r_before_s2
    return r_cascade, t_orig


class Prescription:
    """interfaces[N], media[N-1] (what fills the gap *after* interface i) and
    thicknesses[N-1], stop index, wavelengths (nm)"""

    def __init__(self):
        self.ifcs = []
        self.thi = []
        self.media = []             # ('air',) | ('mirror',) | ('const', n) | ('glass', name) | ('model', nd, vd)
        self.stop = None
        self.wvls = []
        self.ref_wvl = 0
        self.title = ''
        self.private_glasses = {}   # name -> (wavelengths nm, indices): CODE V PRV ... END
        self.unmodelled_types = []  # [(surface index, Zemax TYPE)] imported as their base conic

    # rayoptics/seq/sequential.py:611-668 + rayoptics/elem/transform.py:86-118
    def to_table(self, wvls=None, index_of=None):
        """``index_of(name, wvl_nm)``: see the module docstring.  Default
        :func:`nominal_index`; glasses it does not know are reported
        (:class:`UnknownGlassWarning`) and listed in ``table.fallback_glasses``."""
        default_lookup = index_of is None
        index_of = index_of or nominal_index
        fell_back = []
        wvls = [float(w) for w in (wvls or self.wvls or [550.0])]
        N = len(self.ifcs)
        rows = (abi.Surface * N)()
        n_table = np.ones((len(wvls), N))
        zdir = 1.0
        prev_n = [1.0] * len(wvls)
        for i, s in enumerate(self.ifcs):
            row = rows[i]
            mode = s.mode if 0 < i < N - 1 else 'dummy'     # sequential.py:600-601
            row.mode = abi.MODE_NAMES[mode]
            if s.thinlens_power is not None:                # zmxread.py:327-330, thinlens.py
                row.profile, row.ec = abi.THINLENS, 1.0
                pwr = s.thinlens_power
                ph = row.ph
                ph.kind = abi.PH_HOLOGRAM
                ph.ref_wl = wvls[self.ref_wvl] if self.ref_wvl < len(wvls) else wvls[0]
                ph.a[2] = -1e10
                ph.b[2] = 1. / pwr if pwr != 0 else 1e+10
                ph.flags = 2 if pwr > 0. else 0
            else:
                row.profile = abi.PROFILE_NAMES[s.profile]
                row.cv = float(s.cv)
                if s.profile == 'Spherical':
                    row.cc, row.ec = 0.0, 1.0
                elif s.profile == 'RadialPolynomial':
                    row.ec = float(s.cc) + 1.0 if not hasattr(s, 'ec') else float(s.ec)
                    row.cc = row.ec - 1.0
                else:
                    row.cc = float(s.cc)
                    row.ec = row.cc + 1.0
                if s.profile in ('YToroid', 'XToroid'):
                    row.cR = float(getattr(s, 'cR', 0.0))
                if s.profile in ('EvenPolynomial', 'RadialPolynomial', 'YToroid', 'XToroid'):
                    mnc = 0
                    for k, c in enumerate(s.coefs):
                        if c != 0.0:
                            mnc = k + 1
                    if mnc > abi.MAX_COEF:
                        raise UnsupportedModelError(f'{mnc} asphere coefficients')
                    row.ncoef = mnc
                    for k in range(mnc):
                        row.coefs[k] = float(s.coefs[k])
            if s.phase and s.phase['kind'] == abi.PH_DOE_RADIAL:
                # table._phase_row for a DiffractiveElement (doe.py:225-323)
                if not s.phase.get('radial'):
                    raise UnsupportedModelError('DiffractiveElement without a radial phase function')
                coefs = [float(c) for c in s.phase.get('coefs', [])]
                if len(coefs) > abi.MAX_COEF:
                    raise UnsupportedModelError(f'DiffractiveElement with {len(coefs)} coefficients')
                ph = row.ph
                ph.kind, ph.ncoef = abi.PH_DOE_RADIAL, len(coefs)
                for k, c in enumerate(coefs):
                    ph.coefs[k] = c
                ph.order, ph.ref_wl = float(s.phase.get('order', 1)), float(s.phase.get('ref_wl', 550.))
            elif s.phase:
                ph = row.ph
                ph.kind = s.phase['kind']
                ph.order = float(s.phase.get('order', 1))
                ph.spacing_nm = float(s.phase.get('spacing_nm', 0.0))
                ph.ref_wl = float(s.phase.get('ref_wl', 0.0))
                for k in range(3):
                    ph.a[k] = float(s.phase.get('a', (0., 1., 0.))[k])
            row.max_aperture = float(s.max_aperture)
            if len(s.apertures) > abi.MAX_AP:
                raise UnsupportedModelError('too many clear apertures')
            row.n_ap = len(s.apertures)
            for k, ca in enumerate(s.apertures):
                a = row.ap[k]
                a.is_obscuration = 1 if ca.get('is_obscuration') else 0
                a.x_offset, a.y_offset = float(ca.get('x_offset', 0.)), float(ca.get('y_offset', 0.))
                if ca['kind'] == 'Circular':
                    a.kind, a.a = abi.AP_CIRCULAR, float(ca['radius'])
                elif ca['kind'] == 'Rectangular':
                    a.kind = abi.AP_RECTANGULAR
                    a.a, a.b = float(ca['x_half_width']), float(ca['y_half_width'])
                else:
                    a.kind = abi.AP_ALWAYS_BLOCK
                    a.a, a.b = float(ca.get('x_half_width', 0.)), float(ca.get('y_half_width', 0.))
            # compute_local_transforms (transform.py:79-107) holds r.transpose() -- an
            # F-ordered view -- for every interface but the last, which gets a fresh
            # np.identity(3)
            if i < N - 1:
                r, t = forward_transform(s.decenter, float(self.thi[i]), self.ifcs[i + 1].decenter)
                rt = r.transpose()
            else:
                rt, t = np.identity(3), np.array([0., 0., 0.])
            row.rt_order = rt_order_of(rt)
            for a in range(3):
                for b in range(3):
                    row.rt[3 * a + b] = float(rt[a][b])
                row.t[a] = float(t[a])
            if mode == 'reflect':
                zdir = -zdir
            row.z_dir = zdir
            # index of the gap after this interface, per wavelength
            if i < N - 1:
                m = self.media[i]
                for w, wl in enumerate(wvls):
                    if m[0] == 'air':
                        n = 1.0
                    elif m[0] == 'mirror':
                        n = prev_n[w]
                    elif m[0] == 'const':
                        n = float(m[1])
                    elif m[0] == 'model':
                        if default_lookup:
                            n = model_glass_index(m[1], m[2], wl)
                        elif index_of is reference_fallback_index:
                            n = float(m[1])
                        else:
                            n = float(index_of(f'{m[1]:.6g},{m[2]:.6g}', wl))
                    elif m[1] in self.private_glasses and (default_lookup or
                                                          index_of is reference_fallback_index):
                        # a glass the file defines itself (CODE V PRV ... END: index samples at
                        # the PWL wavelengths); the reference interpolates them with
                        # opticalglass.InterpolatedMedium -- here: linear in wavelength
                        pw, pn = self.private_glasses[m[1]]
                        order = np.argsort(pw)
                        n = float(np.interp(wl, np.asarray(pw)[order], np.asarray(pn)[order]))
                    else:
                        if default_lookup and not knows_glass(m[1]) and m[1] not in fell_back:
                            fell_back.append(m[1])
                        n = float(index_of(m[1], wl))
                    n_table[w, i] = n
                    prev_n[w] = n
        if fell_back:
            import warnings
            warnings.warn('no dispersion data on file for ' + ', '.join(fell_back) +
                          ": traced with n = 1.5 at every wavelength (the reference's "
                          'not-in-catalogue value); pass index_of= to resolve them',
                          UnknownGlassWarning, stacklevel=2)
        tbl = SurfaceTable(rows, n_table, wvls, self.stop)
        tbl.fallback_glasses = tuple(fell_back)
        return tbl


# ---------------------------------------------------------------- .zmx
def _read_text(path):
    for enc in ('utf-16', 'utf-8', 'utf-8-sig', 'iso-8859-1'):   # zmxread.py:60-70
        try:
            with open(path, encoding=enc) as f:
                return f.read()
        except UnicodeError:
            continue
    raise UnsupportedModelError(f'cannot decode {path}')


def read_zmx(path, unknown_types='reference'):
    """Zemax .zmx -> Prescription (rayoptics/zemax/zmxread.py:118-456).

    ``unknown_types``: what to do with a surface TYPE this reader has no model for (the one
    in the reference tree: QED_TYPE, a Forbes Q-type asphere).  'reference' (default) does what
    the reference's importer does -- zmxread.handle_types_and_params (zmxread.py:295-333)
    remembers the type name and otherwise ignores it, so the surface keeps its base conic and
    the XDAT terms are dropped -- but says so: a warning, and ``Prescription.unmodelled_types``
    = [(surface index, type)].  'raise' refuses the file (UnsupportedModelError)."""
    if unknown_types not in ('reference', 'raise'):
        raise ValueError(f'unknown_types={unknown_types!r}')
    p = Prescription()
    cur = -1
    n_clear_ap = 0
    for raw in _read_text(path).splitlines():
        line = raw.strip()
        if not line:
            continue
        parts = line.split(' ', 1)
        cmd = parts[0]
        inputs = parts[1] if len(parts) == 2 else ''
        if cmd == 'NAME':
            p.title = inputs.strip('"')
        elif cmd == 'SURF':
            p.ifcs.append(Ifc())
            p.thi.append(0.0)
            p.media.append(('air',))
            cur = len(p.ifcs) - 1
        elif cmd == 'CURV':
            p.ifcs[cur].cv = float(inputs.split()[0])
        elif cmd == 'DISZ':
            p.thi[cur] = float(inputs)
        elif cmd == 'GLAS':
            it = inputs.split()
            name = it[0]
            if name == 'MIRROR':
                p.ifcs[cur].mode = 'reflect'
                p.media[cur] = ('mirror',)
            elif name == '___BLANK':
                nd, vd = float(it[3]), float(it[4])
                p.media[cur] = ('const', nd) if vd == 0 else ('model', nd, vd)
            elif re.fullmatch(r'[-+]?\d+(\.\d*)?', name) and len(name) == 6:
                p.media[cur] = ('model', 1 + float(name[:3]) / 1000, float(name[3:]) / 10)
            else:
                p.media[cur] = ('glass', name)
        elif cmd == 'STOP':         # SequentialModel.set_stop, sequential.py:306-314
            p.stop = (cur if cur > 0 else 1) if len(p.ifcs) > 2 else None
        elif cmd == 'WAVM':             # WAVM 1 0.55 1
            it = inputs.split()
            w = float(it[1]) * 1e+3
            if w not in p.wvls:
                p.wvls.append(w)
        elif cmd == 'WAVL':
            p.wvls = [float(i) * 1e+3 for i in inputs.split() if i]
        elif cmd == 'TYPE':
            s = p.ifcs[cur]
            typ = inputs.split()[0]
            s.z_type = typ
            if typ == 'EVENASPH':
                s.profile = 'EvenPolynomial'
                s.coefs = [0.0] * 10
            elif typ == 'XOSPHERE':
                s.profile = 'RadialPolynomial'
                s.coefs = []
            elif typ == 'PARAXIAL':
                s.thinlens_power = 0.0
            elif typ == 'DGRATING':
                # DiffractionGrating() defaults: order 1, normal (0,1,0), 1 line/um
                s.phase = dict(kind=abi.PH_GRATING, order=1, a=(0., 1., 0.), spacing_nm=1e6 / 1000.)
            elif typ == 'COORDBRK':         # zmxread.py:318-320
                s.mode = 'phantom'
                s.decenter = new_decenter('decenter')
            elif typ == 'TOROIDAL':         # zmxread.py:308-312 -> YToroid
                s.profile = 'YToroid'
                s.cR = 0.0
                s.coefs = []
            elif typ != 'STANDARD':
                if unknown_types == 'raise':
                    raise UnsupportedModelError(f'Zemax surface type {typ}')
                import warnings
                warnings.warn(f'{os.path.basename(str(path))}: surface {cur} is a {typ}, which is not '
                              'modelled: imported as its base conic, as the reference\'s zmxread does '
                              '(the extra terms are dropped)', stacklevel=2)
                p.unmodelled_types.append((cur, typ))
        elif cmd == 'CONI':
            s = p.ifcs[cur]
            if s.profile == 'Spherical':
                s.profile = 'Conic'
            s.cc = float(inputs.split()[0])
        elif cmd == 'PARM':
            s = p.ifcs[cur]
            i, val = inputs.split()
            i, val = int(i), float(val)
            if s.z_type == 'EVENASPH':
                s.coefs[i - 1] = val
            elif s.z_type == 'PARAXIAL':
                if i == 1:
                    s.thinlens_power = 1 / val
            elif s.z_type == 'DGRATING':
                if i == 1:
                    s.phase['spacing_nm'] = 1e6 / (val * 1000)     # grating_freq_um -> lpmm
                elif i == 2:
                    s.phase['order'] = val
            elif s.z_type == 'TOROIDAL':    # zmxread.py:366-370
                if i == 1:
                    s.cR = 1.0 / val if val != 0.0 else 0.0     # rR setter, profiles.py:1216-1221
                elif i > 1:
                    s.coefs.append(val)
            elif s.z_type == 'COORDBRK':    # zmxread.py:341-355
                if i in (1, 2):
                    s.decenter['dec'][i - 1] = val
                elif i in (3, 4, 5):
                    s.decenter['euler'][i - 3] = val
                elif i == 6 and val != 0:
                    s.decenter['dtype'] = 'reverse'
        elif cmd == 'XDAT':
            s = p.ifcs[cur]
            it = inputs.split()
            if s.z_type == 'XOSPHERE' and int(it[0]) >= 3:
                s.coefs.append(float(it[1]))
        elif cmd == 'DIAM':             # zmxread.py:391-444
            s = p.ifcs[cur]
            it = inputs.split()
            ca_val = float(it[0])
            if ca_val == 0.0:
                ca_val = 1.0
            ca_type = int(it[1])
            if s.thinlens_power is None:
                if not s.apertures:
                    kinds = {0: ('Circular', False), 1: ('Circular', False), 2: ('Circular', True),
                             4: ('Rectangular', False), 5: ('Rectangular', True),
                             6: ('Elliptical', False), 7: ('Elliptical', True)}
                    if ca_type not in kinds:
                        continue
                    kind, obsc = kinds[ca_type]
                    # constructor defaults: surface.py:398-494
                    s.apertures.append(dict(kind=kind, is_obscuration=obsc, radius=1.0,
                                            x_half_width=1.0, y_half_width=1.0))
                    if ca_type in (1, 4, 6):
                        n_clear_ap += 1
                s.apertures[-1]['radius'] = ca_val
            # Surface.set_max_aperture -> ca.set_dimension(max_ap, max_ap), surface.py:174-179
            s.max_aperture = ca_val
            for ca in s.apertures:
                if not ca.get('is_obscuration') or True:
                    ca['radius'] = ca_val
                    ca['x_half_width'] = ca['y_half_width'] = ca_val
        elif cmd == 'OBDC':
            it = inputs.split()
            ca = p.ifcs[cur].apertures[0]
            ca['x_offset'], ca['y_offset'] = float(it[0]), float(it[1])
    # post_process_input, zmxread.py:222-272
    p.thi.pop()
    p.media.pop()
    if math.isinf(p.thi[0]):
        p.thi[0] = 1e10
    if len(p.wvls) > 1 and p.wvls[-1] == 550.0:
        p.wvls.pop()
    p.ref_wvl = len(p.wvls) // 2
    return p


# ---------------------------------------------------------------- .seq
def _is_number(a):
    try:
        float(a)
        return True
    except ValueError:
        return False


_ORDER = {'A': 4, 'B': 6, 'C': 8, 'D': 10, 'E': 12, 'F': 14, 'G': 16, 'H': 18, 'J': 20}


def read_seq(path):
    """CODE V .seq -> Prescription (rayoptics/codev/cmdproc.py:56-99, 164-450).
    Radius mode (RDM) decides whether S/SO/SI carry a radius or a curvature."""
    text = _read_text(path)
    p = Prescription()
    rdm = False
    cur = -1
    in_prv, prv_wvls = False, []
    # '&' continues the command on the next line: the text before the last '&' of a line is
    # joined with the following line, then comments are cut (codev/reader.py:25-38)
    lines, carry = [], None
    for raw in text.splitlines():
        if carry is not None:
            raw = carry + raw
            carry = None
        k = raw.rfind('&')
        if k >= 0:
            carry = raw[:k]
            continue
        lines.append(raw.split('!', 1)[0])
    if carry is not None:
        lines.append(carry.split('!', 1)[0])
    for raw in lines:
        for stmt in raw.split(';'):
            tok = stmt.strip().split()
            if not tok:
                continue
            tla = tok[0].upper()[:3]
            args = tok[1:]
            if in_prv:                      # private catalogue, cmdproc.py:146-158, 358-369
                if tla == 'END':
                    in_prv = False
                elif tla == 'PWL':
                    prv_wvls = [float(a) for a in args]
                elif tok[0][:1] in "'\"" and len(args) == len(prv_wvls) and all(_is_number(a) for a in args):
                    p.private_glasses[tok[0].strip("'\"")] = (prv_wvls, [float(a) for a in args])
                continue
            if tla == 'PRV':
                in_prv, prv_wvls = True, []
            elif tla == 'RDM':
                rdm = not args or args[0].upper() not in ('N', 'NO')
            elif tla == 'TIT':
                p.title = stmt.strip()[3:].strip().strip("'\"")
            elif tla == 'WL':
                p.wvls = [float(a) for a in args]
            elif tla == 'REF':
                p.ref_wvl = int(float(args[0])) - 1
            elif tla in ('S', 'SO', 'SI'):
                s = Ifc()
                p.ifcs.append(s)
                cur = len(p.ifcs) - 1
                v = float(args[0]) if args else 0.0
                s.cv = (1.0 / v if v != 0.0 else 0.0) if rdm else v
                p.thi.append(float(args[1]) if len(args) > 1 else 0.0)
                med = ('air',)
                if len(args) > 2:
                    g = args[2].strip("'\"")        # (the reference's tokenizer drops the quotes)
                    if g.upper() == 'REFL':
                        s.mode = 'reflect'
                        med = ('mirror',)
                    elif _is_number(g):
                        # a fictitious glass code 'nnn.vvv' (cmdproc.py:47-53, 628-638):
                        # n = 1.nnn, v = vv.v; the dispersion model is opticalglass's
                        gc = float(g)
                        ipart = int(gc)
                        mag = int(math.floor(math.log10(ipart))) + 1
                        med = ('model', 1.0 + ipart / 10 ** mag, round(100.0 * (gc - ipart), 6))
                    elif g.upper() != 'AIR':
                        med = ('glass', g)
                p.media.append(med)
            elif tla == 'STO':
                p.stop = cur
            elif tla == 'CON':
                p.ifcs[cur].profile = 'Conic'
            elif tla == 'ASP':
                p.ifcs[cur].profile = 'EvenPolynomial'
                p.ifcs[cur].coefs = [0.0] * 10
            elif tla == 'SPH':
                p.ifcs[cur].profile = 'Spherical'
            elif tla in ('XTO', 'YTO'):         # cmdproc.py:394-397: mutate_profile, nothing else is read
                p.ifcs[cur].profile = 'XToroid' if tla == 'XTO' else 'YToroid'
                p.ifcs[cur].cR = 0.0
                p.ifcs[cur].coefs = []
            elif tla == 'K' and len(tok[0]) == 1:
                p.ifcs[cur].cc = float(args[0])
            elif len(tok[0]) == 1 and tok[0].upper() in _ORDER and p.ifcs and \
                    p.ifcs[cur].profile == 'EvenPolynomial':
                # set_by_order(order, v): coefs[order // 2 - 1] (profiles.py EvenPolynomial)
                p.ifcs[cur].coefs[_ORDER[tok[0].upper()] // 2 - 1] = float(args[0])
            elif tla in ('RDY', 'RDX'):
                v = float(args[0])
                p.ifcs[cur].cv = 1.0 / v if v != 0.0 else 0.0
            elif tla in ('CUY', 'CUX'):
                if _is_number(args[0]):         # with a qualifier it is a solve: ignored (cmdproc.py:380-383)
                    p.ifcs[cur].cv = float(args[0])
            elif tla == 'THI':
                if _is_number(args[0]):         # 'THI HMY 0.0' etc.: a solve, ignored (:384-387)
                    p.thi[cur] = float(args[0])
            elif tla in ('XDE', 'YDE', 'ZDE', 'ADE', 'BDE', 'CDE', 'DAR', 'BEN', 'REV'):
                s = p.ifcs[cur]                 # cmdproc.py:544-576 decenter_data
                if s.decenter is None:
                    s.decenter = new_decenter('decenter')
                if tla in ('XDE', 'YDE', 'ZDE'):
                    s.decenter['dec']['XYZ'.index(tla[0])] = float(args[0])
                elif tla in ('ADE', 'BDE', 'CDE'):
                    s.decenter['euler']['ABC'.index(tla[0])] = float(args[0])
                else:
                    s.decenter['dtype'] = {'DAR': 'dec and return', 'BEN': 'bend',
                                           'REV': 'reverse'}[tla]
            elif tla in ('CIR', 'REX', 'REY', 'ELX', 'ELY'):
                # cmdproc.py:452-497 aperture_data: a new aperture unless the last one is
                # of the same shape; EDG / HOL qualified ones are not clear apertures
                quals = [a.upper() for a in args if not _is_number(a)]
                vals = [float(a) for a in args if _is_number(a)]
                if 'EDG' in quals or 'HOL' in quals or not vals:
                    continue
                kind = {'C': 'Circular', 'R': 'Rectangular', 'E': 'Elliptical'}[tla[0]]
                cas = p.ifcs[cur].apertures
                if not cas or cas[-1]['kind'] != kind:
                    cas.append(dict(kind=kind, is_obscuration=False, radius=1.0,
                                    x_half_width=1.0, y_half_width=1.0))
                ca = cas[-1]
                if 'OBS' in quals:
                    ca['is_obscuration'] = True
                ca[{'R': 'radius', 'X': 'x_half_width', 'Y': 'y_half_width'}[tla[2]]] = vals[0]
            elif tla == 'DIF':                  # cmdproc.py:583-618 diffractive_optic
                if 'DOE' in [a.upper() for a in args]:
                    p.ifcs[cur].phase = dict(kind=abi.PH_DOE_RADIAL, order=1, ref_wl=550., coefs=[],
                                             radial=False)
                elif args:
                    raise UnsupportedModelError(f'.seq diffractive surface DIF {args[0]}')
            elif tla in ('HOR', 'HWL', 'HCT', 'HCO') and p.ifcs[cur].phase and \
                    p.ifcs[cur].phase['kind'] == abi.PH_DOE_RADIAL:
                ph = p.ifcs[cur].phase
                if tla == 'HOR':
                    ph['order'] = float(args[0])
                elif tla == 'HWL':
                    ph['ref_wl'] = float(args[0])
                elif tla == 'HCT':
                    if 'R' in [a.upper() for a in args]:
                        ph['radial'] = True
                else:                           # HCO Cn value
                    cidx = int(args[0][1:])
                    val = float(args[1])
                    coefs = ph['coefs']
                    if cidx <= len(coefs):
                        coefs[cidx - 1] = val
                    elif cidx == len(coefs) + 1:
                        coefs.append(val)
                    else:
                        coefs.extend([0.] * (cidx - len(coefs)))
                        coefs[cidx - 1] = val
            elif tla in ('ADX', 'ADY'):         # cmdproc.py:525-541 aperture_offset
                vals = [float(a) for a in args if _is_number(a)]
                if p.ifcs[cur].apertures and vals:
                    p.ifcs[cur].apertures[-1]['x_offset' if tla == 'ADX' else 'y_offset'] = vals[0]
    p.thi.pop()
    p.media.pop()
    return p


# ---------------------------------------------------------------- .roa
def _roa_medium(m):
    cls = m['__instance_type__'][1]
    at = m.get('attributes', {})
    if cls == 'Air':
        return ('air',)
    if cls == 'ConstantIndex':
        return ('const', at.get('n', at.get('_n', 1.5)))
    if cls in ('ModelGlass', 'InterpolatedMedium'):
        return ('glass', at.get('label', at.get('gname', cls)))
    name = at.get('gname') or at.get('label') or at.get('name') or cls
    return ('glass', name)


def _roa_vec(v, n=3):
    """a NumPy vector as json_tricks writes it: a plain list, or {"__ndarray__": [...], ...}"""
    if isinstance(v, dict):
        v = v.get('__ndarray__', v.get('data'))
    if v is None:
        return [0.0] * n
    out = [float(x) for x in v]
    if len(out) != n:
        raise UnsupportedModelError(f'.roa vector of {len(out)} components where {n} are expected')
    return out


def _roa_decenter(rec):
    """DecenterData of a .roa surface (rayoptics/elem/surface.py:274-337; json_tricks writes
    the instance's __dict__: _dtype (older files: dtype), dec, euler, rot_pt, rot_mat).  Only
    dtype, dec and euler define the transform: rot_mat is derived from euler on update()
    (:312-316) and rot_pt is not read anywhere in the reference."""
    at = rec.get('attributes', rec)
    dtype = at.get('_dtype', at.get('dtype'))
    if dtype not in ('decenter', 'reverse', 'dec and return', 'bend'):
        raise UnsupportedModelError(f'.roa decenter type {dtype!r}')
    d = new_decenter(dtype)
    d['dec'] = _roa_vec(at.get('dec'))
    d['euler'] = _roa_vec(at.get('euler'))
    return d


def read_roa(path):
    """ray-optics .roa (json_tricks dump of the OpticalModel; nesting per SURVEY 8c)"""
    with open(path) as f:
        d = json.load(f)
    om = d['optical_model']['attributes']
    sm = om['seq_model']['attributes']
    prof_dict = om.get('profile_dict', {})
    p = Prescription()
    for ifc in sm['ifcs']:
        at = ifc['attributes']
        cls = ifc['__instance_type__'][1]
        s = Ifc()
        s.mode = at.get('interact_mode', 'transmit')
        s.max_aperture = at.get('max_aperture', 1.0)
        if at.get('decenter') is not None:
            s.decenter = _roa_decenter(at['decenter'])
        if cls == 'ThinLens':
            s.thinlens_power = at.get('_power', at.get('optical_power', 0.0))
        else:
            prof = at.get('profile') or prof_dict[str(at['profile_id'])]
            s.profile = prof['__instance_type__'][1]
            pa = prof['attributes']
            if s.profile not in abi.PROFILE_NAMES:
                raise UnsupportedModelError(f'profile {s.profile}')
            s.cv = pa.get('cv', 0.0)
            s.cc = pa.get('cc', 0.0)
            if 'ec' in pa:
                s.ec = pa['ec']
                s.cc = pa['ec'] - 1.0
            s.coefs = list(pa.get('coefs', []))
        for ca in at.get('clear_apertures', []) or []:
            ca_at = ca['attributes']
            s.apertures.append(dict(kind=ca['__instance_type__'][1],
                                    radius=ca_at.get('radius', 1.0),
                                    x_half_width=ca_at.get('x_half_width', 1.0),
                                    y_half_width=ca_at.get('y_half_width', 1.0),
                                    x_offset=ca_at.get('x_offset', 0.0),
                                    y_offset=ca_at.get('y_offset', 0.0),
                                    is_obscuration=ca_at.get('is_obscuration', False)))
        p.ifcs.append(s)
    for g in sm['gaps']:
        at = g['attributes']
        p.thi.append(at['thi'])
        p.media.append(_roa_medium(at['medium']))
    p.stop = sm.get('stop_surface')
    osp = om.get('optical_spec', {}).get('attributes', {})
    sr = osp.get('spectral_region', osp.get('wvls', {}))
    at = sr.get('attributes', {}) if isinstance(sr, dict) else {}
    p.wvls = list(at.get('wavelengths', []))
    p.ref_wvl = at.get('reference_wvl', 0)
    # a mirror's following gap repeats the medium before it (air stays air)
    return p


def read(path, **kwargs):
    """dispatch on the file extension (keywords go to the format's reader)"""
    ext = str(path).rsplit('.', 1)[-1].lower()
    if ext == 'zmx':
        return read_zmx(path, **kwargs)
    if ext == 'seq':
        return read_seq(path)
    if ext == 'roa':
        return read_roa(path)
    raise UnsupportedModelError(f'unknown prescription format .{ext}')

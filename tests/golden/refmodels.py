"""TEST INFRASTRUCTURE: builds *reference* (rayoptics) OpticalModels for the
golden-vector generator and the ``needs_reference`` differential tests.  Runs
only where /root/reference exists (the build container), never on the GPU box.

Prescription numbers are read from the reference's own data files where they
exist (rayoptics/raytr/tests/ag_dblgauss_s.py, *.roa JSON read with stdlib
``json``); the element/part-tree model is bypassed with
``OpticalModel(do_init=False)`` as described in SURVEY.md section 8c.
"""
import copy
import json
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, '..', '..'))
from oracle import refshim  # noqa: E402

refshim.install()

from rayoptics.optical.opticalmodel import OpticalModel  # noqa: E402
from rayoptics.raytr.opticalspec import PupilSpec, FieldSpec, WvlSpec  # noqa: E402
from rayoptics.elem import profiles, surface  # noqa: E402
from rayoptics.seq import gap  # noqa: E402
from opticalglass.opticalmedium import Air  # noqa: E402
from opticalglass.modelglass import ModelGlass  # noqa: E402

REF_SRC = refshim.REFERENCE_SRC


def new_model(pupil_key, pupil_value, fov_key, fov_value, flds, wvls, ref_wl,
              is_relative=True, obj_thi=1e10):
    opm = OpticalModel(do_init=False)
    sm, osp = opm['seq_model'], opm['optical_spec']
    sm._initialize_arrays()
    osp['pupil'] = PupilSpec(osp, key=list(pupil_key), value=pupil_value)
    osp['fov'] = FieldSpec(osp, key=list(fov_key), value=fov_value,
                           flds=list(flds), is_relative=is_relative)
    osp['wvls'] = WvlSpec(list(wvls), ref_wl=ref_wl)
    sm.gaps[0].thi = obj_thi
    return opm


def finish(opm, do_apertures=True):
    sm, osp = opm['seq_model'], opm['optical_spec']
    sm.do_apertures = do_apertures
    sm.update_model()
    osp.update_model()
    opm.update_optical_properties()
    return opm


def dblgauss(obj_thi=None, vig=0.4):
    """13-interface double Gauss, rayoptics/raytr/tests/ag_dblgauss_s.py;
    spec from rayoptics/codev/tests/ag_dblgauss.seq (EPD 50, fields 0/10/14
    deg, 656.3/587.6/486.1 nm).  Config C2 of BASELINE.json."""
    sys.path.insert(0, os.path.join(REF_SRC, 'rayoptics', 'raytr', 'tests'))
    import ag_dblgauss_s as dblg
    d = copy.deepcopy(dblg.ag_dblgauss)
    opm = new_model(('object', 'epd'), 50.0, ('object', 'angle'), 14.0,
                    [0., 10. / 14., 1.0],
                    [(656.3, 1.0), (587.6, 2.0), (486.1, 1.0)], 1,
                    obj_thi=d[0][1] if obj_thi is None else obj_thi)
    sm = opm['seq_model']
    for i, row in enumerate(d[1:-1]):
        sm.add_surface([row[0], row[1], row[2], row[3]] if row[2] != 1
                       else [row[0], row[1]])
        if i == 5:
            sm.set_stop()
    finish(opm)
    f = opm['optical_spec']['fov'].fields
    f[1].vuy = f[1].vly = vig / 2
    f[2].vuy = f[2].vly = vig
    return opm


def singlet():
    """4-interface N-BK7-like singlet at finite conjugates (the shape of
    rayoptics/models/singlet_f5.roa).  Config C1 of BASELINE.json."""
    opm = new_model(('object', 'epd'), 10.0, ('object', 'height'), 5.0,
                    [0., 1.0], [(650.0, 1.0)], 0, obj_thi=100.0)
    sm = opm['seq_model']
    sm.add_surface([1 / 51.0, 4.0, 1.5168, 64.17])
    sm.set_stop()
    sm.add_surface([-1 / 51.0, 96.0])
    return finish(opm)


def rc_telescope(field_stop=True):
    """Ritchey-Chretien pair (rayoptics/models/Ritchey_Chretien.roa: two
    Conic mirrors, thi = -22, z_dir = [1,-1,1]) plus a hand-added dummy field
    stop with a Circular clear aperture.  Config C4 of BASELINE.json."""
    with open(os.path.join(REF_SRC, 'rayoptics', 'models',
                           'Ritchey_Chretien.roa')) as f:
        om = json.load(f)['optical_model']['attributes']
    sm_j = om['seq_model']['attributes']
    pd = om['profile_dict']
    prof = [pd[i['attributes']['profile_id']]['attributes'] for i in sm_j['ifcs']]
    thi = [g['attributes']['thi'] for g in sm_j['gaps']]
    opm = new_model(('object', 'epd'), 7.5, ('object', 'angle'), 0.35,
                    [0., 0.25, 0.5, 0.75, 1.0], [(550.0, 1.0)], 0,
                    obj_thi=thi[0])
    sm = opm['seq_model']
    sm.add_surface([prof[1]['cv'], thi[1], 'REFL'])
    sm.ifcs[sm.cur_surface].profile = profiles.Conic(c=prof[1]['cv'], cc=prof[1]['cc'])
    sm.set_stop()
    back = thi[2]
    sm.add_surface([prof[2]['cv'], back - 2.0 if field_stop else back, 'REFL'])
    sm.ifcs[sm.cur_surface].profile = profiles.Conic(c=prof[2]['cv'], cc=prof[2]['cc'])
    if field_stop:
        sm.add_surface([0.0, 2.0])
        fs = sm.ifcs[sm.cur_surface]
        fs.interact_mode = 'dummy'
    finish(opm)
    if field_stop:
        # after update (set_clear_apertures would resize it otherwise)
        fs.clear_apertures = [surface.Circular(radius=0.42)]
    return opm


def _medium_from_json(m):
    kind = m['__instance_type__'][1]
    a = m.get('attributes', {})
    if kind == 'Air':
        return Air()
    if kind == 'ModelGlass':
        return ModelGlass(a['n'], a['v'], a.get('label', ''))
    if kind == 'ConstantIndex':
        from opticalglass.opticalmedium import ConstantIndex
        return ConstantIndex(a['n'], a.get('label', ''))
    raise ValueError(f'medium {kind} needs a glass catalog')


def _profile_from_json(p):
    kind = p['__instance_type__'][1]
    a = p['attributes']
    if kind == 'Spherical':
        return profiles.Spherical(c=a['cv'])
    if kind == 'Conic':
        return profiles.Conic(c=a['cv'], cc=a['cc'])
    if kind == 'EvenPolynomial':
        return profiles.EvenPolynomial(c=a['cv'], cc=a['cc'], coefs=list(a['coefs']))
    if kind == 'RadialPolynomial':
        return profiles.RadialPolynomial(c=a['cv'], ec=a['ec'], coefs=list(a['coefs']))
    raise ValueError(kind)


def load_roa(path, pupil=None, fov=None, flds=None, wvls=None, ref_wl=None,
             is_relative=False):
    """.roa (json_tricks) -> reference model, with stdlib json only.
    Profiles, gaps, z_dir and stop are taken from the file; the optical spec
    comes from the file unless overridden."""
    with open(path) as f:
        om = json.load(f)['optical_model']['attributes']
    sm_j = om['seq_model']['attributes']
    pd = om.get('profile_dict', {})
    osp_j = om['optical_spec']['attributes']
    sr = osp_j['spectral_region']['attributes']
    if wvls is None:
        wvls = list(zip(sr['wavelengths'], sr['spectral_wts']))
        ref_wl = sr['reference_wvl']
    pj = osp_j['pupil']['attributes']
    pkey = pupil[0] if pupil else tuple(pj['_key'][1:]) if '_key' in pj else tuple(pj['key'][1:])
    pval = pupil[1] if pupil else pj['value']
    fj = osp_j['field_of_view']['attributes']
    fkey = fov[0] if fov else tuple((fj.get('_key') or fj['key'])[1:])
    fval = fov[1] if fov else fj['value']
    if flds is None:
        flds = [fl['attributes']['y'] for fl in fj['fields']]
        is_relative = fj.get('is_relative', False)
    thi = [g['attributes']['thi'] for g in sm_j['gaps']]
    opm = new_model(pkey, pval, fkey, fval, flds, wvls, ref_wl,
                    is_relative=is_relative, obj_thi=thi[0])
    sm = opm['seq_model']
    ifcs_j = sm_j['ifcs']
    for k in range(1, len(ifcs_j) - 1):
        a = ifcs_j[k]['attributes']
        pj_ = a.get('profile') or pd[a['profile_id']]
        s = surface.Surface(profile=_profile_from_json(pj_),
                            interact_mode=a['interact_mode'],
                            max_ap=a['max_aperture'])
        dj = a.get('decenter')
        if dj is not None:          # DecenterData as json_tricks writes it (the instance's __dict__)
            da = dj['attributes']
            vec = lambda v: v['__ndarray__'] if isinstance(v, dict) else v   # noqa: E731
            dec, eul = vec(da['dec']), vec(da['euler'])
            s.decenter = surface.DecenterData(da.get('_dtype', da.get('dtype')), x=dec[0], y=dec[1],
                                              alpha=eul[0], beta=eul[1], gamma=eul[2])
            s.decenter.dec[2] = dec[2]
        for cj in a.get('clear_apertures') or []:
            ca = cj['attributes']
            kind = cj['__instance_type__'][1]
            kw = dict(x_offset=ca.get('x_offset', 0.0), y_offset=ca.get('y_offset', 0.0),
                      is_obscuration=ca.get('is_obscuration', False))
            if kind == 'Circular':
                s.clear_apertures.append(surface.Circular(radius=ca['radius'], **kw))
            else:
                s.clear_apertures.append(getattr(surface, kind)(
                    x_half_width=ca['x_half_width'], y_half_width=ca['y_half_width'], **kw))
        g = gap.Gap(thi[k], _medium_from_json(sm_j['gaps'][k]['attributes']['medium']))
        sm.insert(s, g, z_dir=sm_j['z_dir'][k] if 'z_dir' in sm_j else 1)
        if k == sm_j['stop_surface']:
            sm.set_stop()
    return finish(opm, do_apertures=False)


def nikkor():
    """29-interface zoom with 4 EvenPolynomial aspheres
    (rayoptics/optical/tests/Nikon Nikkor Z 14-30mm f-4 S.roa); ModelGlass
    (n, v) media from the file.  Stand-in for config C3; narrower fields than
    the file's 57.7 deg so that the non-wide-angle ray start applies."""
    path = os.path.join(REF_SRC, 'rayoptics', 'optical', 'tests',
                        'Nikon Nikkor Z 14-30mm f-4 S.roa')
    return load_roa(path, fov=(('object', 'angle'), 30.0), flds=[0., 15., 30.],
                    is_relative=False)


def cell_phone():
    """13-interface cell-phone camera, 8 RadialPolynomial aspheres
    (rayoptics/optical/tests/cell_phone_camera.roa) -- the reference's only
    timed asphere model (rayoptics/raytr/tests/trace_results.txt:10)."""
    path = os.path.join(REF_SRC, 'rayoptics', 'optical', 'tests',
                        'cell_phone_camera.roa')
    return load_roa(path)


def tilted_singlet():
    """synthetic decentered/tilted system exercising general (R^T, t), a
    phantom coordinate break, a Rectangular aperture and a Circular
    obscuration."""
    opm = new_model(('object', 'epd'), 8.0, ('object', 'angle'), 2.0,
                    [0., 1.0], [(550.0, 1.0), (650.0, 1.0)], 0, obj_thi=200.0)
    sm = opm['seq_model']
    sm.add_surface([0.02, 5.0, 1.62, 36.0])
    sm.set_stop()
    sm.ifcs[sm.cur_surface].decenter = surface.DecenterData(
        'dec and return', x=0.3, y=-0.2, alpha=3.0, beta=-2.0, gamma=10.0)
    sm.add_surface([-0.015, 10.0])
    sm.add_coord_break(5.0, decenter_data=surface.DecenterData(
        'decenter', x=0.1, y=0.4, alpha=-4.0, beta=1.5))
    sm.add_surface([0.01, 3.0, 1.5, 60.0])
    sm.ifcs[sm.cur_surface].profile = profiles.Conic(c=0.01, cc=-0.7)
    sm.add_surface([0.0, 60.0])
    finish(opm, do_apertures=False)
    for i, ifc in enumerate(sm.ifcs):
        ifc.max_aperture = 12.0
    sm.ifcs[2].clear_apertures = [surface.Rectangular(x_half_width=5.0,
                                                       y_half_width=4.0,
                                                       x_offset=0.25)]
    sm.ifcs[4].clear_apertures = [surface.Circular(radius=6.0),
                                  surface.Circular(radius=0.8, y_offset=0.5,
                                                   is_obscuration=True)]
    return opm


def toroid_lens():
    """synthetic anamorphic pair: a YToroid and an XToroid surface (aspheric
    toroids on a conic base, rayoptics/elem/profiles.py:1117-1437), both
    intersected by the Spencer-Murty Newton iteration"""
    opm = new_model(('object', 'epd'), 6.0, ('object', 'angle'), 3.0,
                    [0., 1.0], [(550.0, 1.0), (450.0, 1.0)], 0, obj_thi=150.0)
    sm = opm['seq_model']
    sm.add_surface([0.02, 4.0, 1.58, 41.0])
    sm.set_stop()
    sm.ifcs[sm.cur_surface].profile = profiles.YToroid(
        c=0.025, cR=0.012, cc=-0.3, coefs=[0., 2.0e-5, -3.0e-7, 0., 0., 0., 0., 0., 0., 0.])
    sm.add_surface([-0.01, 6.0])
    sm.add_surface([0.015, 3.0, 1.49, 70.0])
    sm.ifcs[sm.cur_surface].profile = profiles.XToroid(
        c=0.018, cR=-0.006, cc=0.4, coefs=[1.0e-4, -1.5e-5, 0., 0., 0., 0., 0., 0., 0., 0.])
    sm.add_surface([-0.02, 55.0])
    finish(opm, do_apertures=False)
    for ifc in sm.ifcs:
        ifc.max_aperture = 9.0
    return opm


def litho_c5():
    """BASELINE.json configs[4] stand-in: rayoptics/zemax/tests/US05831776-1.zmx
    (44 interfaces) through the reference's own Zemax importer, 9 fields x 5
    wavelengths; the catalogue glass (SILICA, unknown to the stubbed catalogue)
    gets the Malitson dispersion formula so that the wavelengths differ."""
    import pathlib
    from rayoptics.zemax import zmxread
    from rayoptics_amd import ingest
    path = pathlib.Path(REF_SRC) / 'rayoptics' / 'zemax' / 'tests' / 'US05831776-1.zmx'
    opm, _info = zmxread.read_lens(None, path.open(encoding='utf-8').read(), do_update=False)
    sm, osp = opm['seq_model'], opm['optical_spec']

    class Silica(refshim.OpticalMedium):
        def __init__(self):
            super().__init__(1.5084, 'SILICA', 'nominal')

        def rindex(self, w):
            return ingest.sellmeier_index('SILICA', refshim.get_wavelength(w))
    for g in sm.gaps:
        if g.medium.name().startswith('not '):
            g.medium = Silica()
    osp['wvls'] = WvlSpec([(247.8, 1.), (247.9, 1.), (248.0, 1.), (248.1, 1.), (248.2, 1.)], ref_wl=2)
    osp['fov'] = FieldSpec(osp, key=list(osp['fov'].key), value=12.0,
                           flds=[i / 8 for i in range(9)], is_relative=True)
    return finish(opm, do_apertures=False)


def zmx_evenasph_c3():
    """BASELINE.json configs[2]: "Zemax .zmx import, 12-surface even-asphere zoom, 3 fields
    x 3 wvls" -- rayoptics/zemax/tests/US08427765-1.ZMX (13 interfaces, one EVENASPH, three
    'real height' image fields, F/d/C lines, image f/2.1) through the reference's own Zemax
    importer.  Its five glasses (J-LAK14, L-TIM28, SF11, TAF3, TAFD30) are unknown to the
    stubbed catalogue; they get their nominal catalogue (nd, vd) through the test ModelGlass
    so that the system is the lens the file describes and the wavelengths differ."""
    import pathlib
    from rayoptics.zemax import zmxread
    path = pathlib.Path(REF_SRC) / 'rayoptics' / 'zemax' / 'tests' / 'US08427765-1.ZMX'
    opm, _info = zmxread.read_lens(None, path.open(encoding='utf-8').read(), do_update=False)
    nominal = {'J-LAK14': (1.69680, 55.46), 'L-TIM28_MOLD': (1.68893, 31.08),
               'SF11': (1.78472, 25.76), 'TAF3': (1.80420, 46.50), 'TAFD30': (1.88300, 40.80)}
    for g in opm['seq_model'].gaps:
        name = g.medium.name()
        if name.startswith('not '):
            nd, vd = nominal[name[4:]]
            g.medium = ModelGlass(nd, vd, name[4:])
    return finish(opm, do_apertures=False)


def telecentric():
    """image-space telecentric singlet (stop at the front focal plane): the
    reference sphere of the axial field is 'kinda big' (waveabr.py:213-216), so the
    OPD goes through wave_abr_full_calc_inf_ref (waveabr.py:356-424)"""
    opm = new_model(('object', 'epd'), 6.0, ('object', 'angle'), 3.0, [0., 0.7, 1.0],
                    [(486.1, 1.0), (550.0, 1.0)], 1, obj_thi=1e10)
    sm = opm['seq_model']
    sm.add_surface([0.0, 32.38806981225028])
    sm.set_stop()
    sm.add_surface([1 / 40.0, 5.0, 1.6, 50.0])
    sm.add_surface([-1 / 40.0, 30.0])
    return finish(opm)


# ---- the models of the reference's own benchmark (rayoptics/raytr/tests/time_trace.py) ----------
class NominalGlass(refshim.OpticalMedium):
    """a named catalogue glass with the dispersion formula the product's ingest has on file
    (rayoptics_amd.ingest.nominal_index); stands in for opticalglass, which is absent here"""

    def __init__(self, name):
        from rayoptics_amd import ingest
        self._name = name
        super().__init__(ingest.nominal_index(name, 587.6), name, 'nominal')

    def rindex(self, w):
        from rayoptics_amd import ingest
        return ingest.nominal_index(self._name, refshim.get_wavelength(w))


def _nominal_media(opm):
    """glasses the (stubbed, empty) catalogue did not find -- ConstantIndex(1.5, 'not NAME'),
    as the reference's importers leave them -- get their nominal dispersion"""
    for g in opm['seq_model'].gaps:
        nm = g.medium.name()
        if nm.startswith('not '):
            g.medium = NominalGlass(nm[4:].strip())
    return opm


def time_trace_model(rel):
    """one of the ten models rayoptics/raytr/tests/time_trace.py times (paths relative to the
    rayoptics package): .seq through the reference's CODE V importer, .roa through load_roa"""
    import pathlib
    path = pathlib.Path(REF_SRC) / 'rayoptics' / rel
    if rel.endswith('.seq'):
        from rayoptics.codev import cmdproc
        from rayoptics.seq.sequential import SequentialModel
        from rayoptics.optical.opticalmodel import OpticalModel
        saved = SequentialModel.set_clear_apertures, OpticalModel.update_model
        SequentialModel.set_clear_apertures = lambda self, **kw: None
        OpticalModel.update_model = lambda self, **kw: None
        try:
            opm, _info = cmdproc.read_lens(path, do_update=False)
        finally:
            SequentialModel.set_clear_apertures, OpticalModel.update_model = saved
        return finish(_nominal_media(opm), do_apertures=True)
    global _medium_from_json
    plain = _medium_from_json

    def with_glasses(m):
        try:
            return plain(m)
        except ValueError:
            a = m.get('attributes', {})
            return NominalGlass(a.get('gname') or a.get('label') or a.get('name'))
    _medium_from_json = with_glasses
    try:
        return load_roa(str(path))
    finally:
        _medium_from_json = plain


TIME_TRACE_MODELS = [       # (workload name, file, row of trace_results.txt, published rays/s)
    ('tt_singlet_seq', 'codev/tests/singlet.seq', 'singlet', 7955),
    ('tt_landscape', 'codev/tests/landscape_lens.seq', 'landscape lens', 6197),
    ('tt_triplet', 'models/Sasian Triplet.roa', 'Sasian triplet', 3740),
    ('tt_two_sph_mirrors', 'models/TwoSphericalMirror.roa', '2 spherical mirrors (spheres)', 8347),
    ('tt_two_mirrors_conic', 'models/TwoMirror.roa', '2 spherical mirrors (conics)', 7994),
    ('tt_paraboloid', 'codev/tests/paraboloid.seq', 'paraboloid', 7957),
    ('tt_cassegrain', 'models/Cassegrain.roa', 'Cassegrain', 7973),
]

"""f3 -- prescription ingest without the reference's object model: tables
parsed straight from .zmx / .seq / .roa files (rayoptics_amd.ingest, stdlib
only) must equal, field by field, the tables ``SurfaceTable.from_seq_model``
extracts from the model the reference's OWN importer builds from the same file
(rayoptics/zemax/zmxread.py, rayoptics/codev/cmdproc.py; .roa files through the
reference classes, tests/golden/refmodels.load_roa).  Build container only.

Glass names resolve to n = 1.5 on both sides (the reference's behaviour when
its catalogue does not know a glass, rayoptics/seq/medium.py:172-203; the glass
catalogue package itself is absent here: catalogue parity stays unpinned)."""
import os
import pathlib

import numpy as np
import pytest

REF = '/root/reference/src/rayoptics'


def rows_equal(a, b, what):
    from rayoptics_amd import abi
    assert a.n_ifcs == b.n_ifcs, what
    for i, (ra, rb) in enumerate(zip(a.rows, b.rows)):
        for name, _t in abi.Surface._fields_:
            va, vb = getattr(ra, name), getattr(rb, name)
            if name in ('ap', 'ph'):
                assert bytes(va) == bytes(vb), (what, i, name)
            elif hasattr(va, '__len__'):
                assert list(va) == list(vb), (what, i, name, list(va), list(vb))
            else:
                assert va == vb, (what, i, name, va, vb)
    np.testing.assert_array_equal(a.n_table, b.n_table, err_msg=what)
    assert a.wvls == b.wvls and a.stop_idx == b.stop_idx, (what, a.wvls, b.wvls, a.stop_idx, b.stop_idx)


@pytest.mark.needs_reference
@pytest.mark.parametrize('rel', ['zemax/tests/US05831776-1.zmx', 'zemax/tests/354710-C-Zemax(ZMX).zmx',
                                 'elem/tests/ACL3026U-Zemax(ZMX).zmx',
                                 'zemax/tests/zmax_37992.zmx',          # a COORDBRK fold
                                 'zemax/tests/HoO-V2C18Ex46.zmx'])      # 5 COORDBRK + a TOROIDAL
def test_zmx_table_equals_reference_import(rel):
    from oracle import refshim
    refshim.install()
    from rayoptics.zemax import zmxread
    from rayoptics_amd import SurfaceTable, ingest
    path = pathlib.Path(REF) / rel
    for enc in ('utf-16', 'utf-8', 'iso-8859-1'):
        try:
            inpt = path.open(encoding=enc).read()
            break
        except UnicodeError:
            pass
    opm, _info = zmxread.read_lens(None, inpt, do_update=False)     # the reference's importer
    opm['seq_model'].update_model()
    theirs = SurfaceTable.from_seq_model(opm['seq_model'])
    ours = ingest.read_zmx(str(path)).to_table(index_of=ingest.reference_fallback_index)
    rows_equal(ours, theirs, rel)


@pytest.mark.needs_reference
@pytest.mark.parametrize('rel', ['codev/tests/ag_dblgauss.seq', 'codev/tests/rc_f16.seq',
                                 'codev/tests/singlet.seq',
                                 'codev/tests/threemir.seq',            # DAR decenters + tilts, REX/REY/ADY
                                 'codev/tests/CODV_35571.seq'])         # off-axis parabola, XDE..CDE, CIR
def test_seq_table_equals_reference_import(rel):
    from oracle import refshim
    refshim.install()
    from rayoptics.codev import cmdproc
    from rayoptics_amd import SurfaceTable, ingest
    path = pathlib.Path(REF) / rel
    # files that carry aperture data make read_lens trace rays to size the *other* surfaces
    # (set_clear_apertures, cmdproc.py:89-95: control plane, it needs the element model);
    # max_aperture is neutralised below, so that step is skipped
    from rayoptics.seq.sequential import SequentialModel
    from rayoptics.optical.opticalmodel import OpticalModel
    saved = SequentialModel.set_clear_apertures, OpticalModel.update_model
    SequentialModel.set_clear_apertures = lambda self, **kw: None
    OpticalModel.update_model = lambda self, **kw: None
    try:
        opm, _info = cmdproc.read_lens(path, do_update=False)
    finally:
        SequentialModel.set_clear_apertures, OpticalModel.update_model = saved
    sm = opm['seq_model']
    sm.update_model()
    theirs = SurfaceTable.from_seq_model(sm)
    ours = ingest.read_seq(str(path)).to_table(index_of=ingest.reference_fallback_index)
    # CODE V listings give max_aperture no value: the reference derives it later
    # from traced rays (set_clear_apertures, control plane); compare with it neutralised
    for t in (ours, theirs):
        for r in t.rows:
            r.max_aperture = 1.0
    rows_equal(ours, theirs, rel)


@pytest.mark.needs_reference
@pytest.mark.parametrize('rel', ['models/Ritchey_Chretien.roa', 'optical/tests/cell_phone_camera.roa',
                                 'optical/tests/Nikon Nikkor Z 14-30mm f-4 S.roa'])
def test_roa_table_equals_reference_model(rel):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import refmodels as rm
    from rayoptics_amd import SurfaceTable, ingest
    path = os.path.join(REF, rel)
    opm = rm.load_roa(path)
    sm = opm['seq_model']
    theirs = SurfaceTable.from_seq_model(sm)
    wvls = theirs.wvls
    idx = {}
    for i, g in enumerate(sm.gaps):                 # the media the reference model holds
        idx[i] = [g.medium.rindex(w) for w in wvls]
    pres = ingest.read_roa(path)
    ours = pres.to_table(wvls=wvls, index_of=ingest.reference_fallback_index)
    # refractive indices: the .roa names catalogue glasses; both sides take the
    # evaluated numbers of the reference model (catalogue parity is out of scope)
    ours.n_table[:, :len(sm.gaps)] = np.array([idx[i] for i in range(len(sm.gaps))]).T
    ours.n_table[:, len(sm.gaps):] = theirs.n_table[:, len(sm.gaps):]
    # (load_roa gives the object and image surfaces the class default aperture)
    for t in (ours, theirs):
        t.rows[0].max_aperture = t.rows[-1].max_aperture = 1.0
    rows_equal(ours, theirs, rel)


def test_nominal_dispersion_matches_the_codev_listing():
    """SELLMEIER entries of the double Gauss glasses against the indices CODE V
    printed for them (rayoptics/codev/tests/ag_dblgauss.lis:30-34, 6 decimals)"""
    from rayoptics_amd import ingest
    lis = {'NSSK2_SCHOTT': (1.618769, 1.622292, 1.630455), 'NSK2_SCHOTT': (1.604134, 1.607379, 1.614860),
           'F5_SCHOTT': (1.598744, 1.603417, 1.614617), 'NSK16_SCHOTT': (1.617271, 1.620408, 1.627559)}
    for name, ns in lis.items():
        for w, n in zip((656.3, 587.6, 486.1), ns):
            assert abs(ingest.nominal_index(name, w) - n) < 2e-5, (name, w)
    assert abs(ingest.nominal_index('N-BK7', 587.5618) - 1.5168) < 1e-5
    assert abs(ingest.nominal_index('SILICA', 248.0) - 1.5084) < 5e-4
    assert ingest.nominal_index('UNOBTAINIUM', 500.) == 1.5


@pytest.mark.needs_reference
@pytest.mark.parametrize('rel', ['codev/tests/threemir.seq', 'codev/tests/CODV_35571.seq'])
def test_decentered_ingest_traces_like_the_reference(rel):
    """end to end on a decentered, tilted system: the table parsed from the file, traced
    by the oracle, against the reference's rt.trace on the model its own importer built"""
    from oracle import refshim
    refshim.install()
    from rayoptics.codev import cmdproc
    from rayoptics.seq.sequential import SequentialModel
    from rayoptics.optical.opticalmodel import OpticalModel
    import rayoptics.raytr.raytrace as rt
    from rayoptics.raytr.traceerror import TraceError
    from oracle import oracle
    from rayoptics_amd import abi, ingest
    path = pathlib.Path(REF) / rel
    saved = SequentialModel.set_clear_apertures, OpticalModel.update_model
    SequentialModel.set_clear_apertures = lambda self, **kw: None
    OpticalModel.update_model = lambda self, **kw: None
    try:
        opm, _info = cmdproc.read_lens(path, do_update=False)
    finally:
        SequentialModel.set_clear_apertures, OpticalModel.update_model = saved
    sm = opm['seq_model']
    sm.update_model()
    tbl = ingest.read_seq(str(path)).to_table(index_of=ingest.reference_fallback_index)
    N = tbl.n_ifcs
    wvl = tbl.wvls[0]
    rng = np.random.default_rng(5)
    R = 40
    tgt = np.stack([rng.uniform(-10, 10, R), rng.uniform(-10, 10, R), np.full(R, 1e3)])
    pt0 = np.zeros((3, R))
    pt0[2] = tbl.rows[0].t[2] - 1e3
    d0 = tgt / np.linalg.norm(tgt, axis=0)
    opts = oracle.make_opts(flags=abi.INTERSECT_OBJ, first_surf=1, last_surf=N - 2)
    orc = oracle.trace_rays(tbl, pt0, d0, 0, opts)
    n_ok = 0
    for r in range(R):
        try:
            ray, op, _w = rt.trace(sm, pt0[:, r].copy(), d0[:, r].copy(), wvl)
        except TraceError:
            assert orc.status[r] != abi.OK
            continue
        n_ok += 1
        assert orc.status[r] == abi.OK and op == orc.op[r]
        for k, seg in enumerate(ray):
            np.testing.assert_array_equal(seg[0], orc.seg[k, 0:3, r])
            np.testing.assert_array_equal(seg[1], orc.seg[k, 3:6, r])
            assert seg[2] == orc.seg[k, 6, r]
            np.testing.assert_array_equal(seg[3], orc.seg[k, 7:10, r])
    assert n_ok > 20


SYNTHETIC_ZMX = """VERS 140124 258 36214
MODE SEQ
NAME synthetic: every surface type the reader knows
UNIT MM X W X CM MR CPMM
ENPD 8.0
WAVM 1 0.4861 1
WAVM 2 0.5876 1
WAVM 3 0.6563 1
PWAV 2
SURF 0
  TYPE STANDARD
  CURV 0.0 0 0 0 0 ""
  DISZ INFINITY
  DIAM 0 0 0 0 1 ""
SURF 1
  TYPE STANDARD
  CURV 0.02 0 0 0 0 ""
  DISZ 3.0
  GLAS ___BLANK 1 0 1.5168 0 0 0 0 0 0
  DIAM 6.0 1 0 0 1 ""
  OBDC 0.25 -0.5
SURF 2
  STOP
  TYPE XOSPHERE
  CURV -0.01 0 0 0 0 ""
  CONI -0.3
  XDAT 1 4
  XDAT 2 1.0
  XDAT 3 0.0
  XDAT 4 1.5e-05
  XDAT 5 0.0
  XDAT 6 -2.0e-08
  DISZ 5.0
  DIAM 5.5 0 0 0 1 ""
SURF 3
  TYPE PARAXIAL
  CURV 0.0 0 0 0 0 ""
  PARM 1 80.0
  DISZ 10.0
  DIAM 5.0 0 0 0 1 ""
SURF 4
  TYPE DGRATING
  CURV 0.0 0 0 0 0 ""
  PARM 1 0.3
  PARM 2 -1
  DISZ 20.0
  DIAM 5.0 0 0 0 1 ""
SURF 5
  TYPE STANDARD
  CURV 0.0 0 0 0 0 ""
  GLAS MIRROR 0 0 1.5 40.0
  DISZ -15.0
  DIAM 7.0 4 0 0 1 ""
SURF 6
  TYPE EVENASPH
  CURV 0.005 0 0 0 0 ""
  PARM 1 0.0
  PARM 2 1.0e-05
  PARM 3 0.0
  PARM 4 0.0
  PARM 5 0.0
  PARM 6 0.0
  PARM 7 0.0
  PARM 8 0.0
  DISZ -4.0
  DIAM 6.0 2 0 0 1 ""
SURF 7
  TYPE STANDARD
  CURV 0.0 0 0 0 0 ""
  DISZ 0.0
  DIAM 3.0 0 0 0 1 ""
"""


@pytest.mark.needs_reference
def test_synthetic_zmx_with_every_surface_type(tmp_path):
    """no file of the reference tree carries XOSPHERE, PARAXIAL, DGRATING, an offset or
    obscuring aperture: a synthetic .zmx with all of them, through both importers"""
    from oracle import refshim
    refshim.install()
    from rayoptics.zemax import zmxread
    from rayoptics_amd import SurfaceTable, ingest
    path = tmp_path / 'synthetic.zmx'
    path.write_text(SYNTHETIC_ZMX, encoding='utf-8')
    opm, _info = zmxread.read_lens(None, SYNTHETIC_ZMX, do_update=False)
    opm['seq_model'].update_model()
    theirs = SurfaceTable.from_seq_model(opm['seq_model'])
    ours = ingest.read_zmx(str(path)).to_table(index_of=ingest.reference_fallback_index)
    rows_equal(ours, theirs, 'synthetic.zmx')
    from rayoptics_amd import abi
    kinds = [r.profile for r in ours.rows]
    assert abi.PROFILE_NAMES['RadialPolynomial'] in kinds and abi.THINLENS in kinds
    assert ours.rows[4].ph.kind == abi.PH_GRATING and ours.rows[5].mode == abi.MODE_NAMES['reflect']


@pytest.mark.needs_reference
def test_synthetic_zmx_traces_like_the_reference(tmp_path):
    """... and the table parsed from that file, traced by the oracle, against the reference's
    rt.trace on the model its own importer built: thin lens, grating, radial and even aspheres,
    a mirror, offset / rectangular / obscuring apertures in one path, three wavelengths"""
    from oracle import refshim
    refshim.install()
    from rayoptics.zemax import zmxread
    import rayoptics.raytr.raytrace as rt
    from rayoptics.raytr.traceerror import TraceError
    from oracle import oracle
    from rayoptics_amd import abi, ingest
    path = tmp_path / 'synthetic.zmx'
    path.write_text(SYNTHETIC_ZMX, encoding='utf-8')
    opm, _info = zmxread.read_lens(None, SYNTHETIC_ZMX, do_update=False)
    sm = opm['seq_model']
    sm.update_model()
    tbl = ingest.read_zmx(str(path)).to_table(index_of=ingest.reference_fallback_index)
    N = tbl.n_ifcs
    rng = np.random.default_rng(12)
    R = 60
    tgt = np.stack([rng.uniform(-5, 5, R), rng.uniform(-5, 5, R), np.full(R, 1e3)])
    pt0 = np.zeros((3, R))
    pt0[2] = tbl.rows[0].t[2] - 1e3
    d0 = tgt / np.linalg.norm(tgt, axis=0)
    n_ok = n_err = 0
    for wi, wvl in enumerate(tbl.wvls):
        opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES, first_surf=1, last_surf=N - 2)
        orc = oracle.trace_rays(tbl, pt0, d0, wi, opts)
        for r in range(R):
            try:
                ray, op, _w = rt.trace(sm, pt0[:, r].copy(), d0[:, r].copy(), wvl, check_apertures=True)
            except TraceError as e:
                n_err += 1
                assert orc.status[r] != abi.OK and orc.fail_surf[r] == e.surf
                continue
            n_ok += 1
            assert orc.status[r] == abi.OK and op == orc.op[r]
            for k, seg in enumerate(ray):
                np.testing.assert_array_equal(seg[0], orc.seg[k, 0:3, r])
                np.testing.assert_array_equal(seg[1], orc.seg[k, 3:6, r])
                assert seg[2] == orc.seg[k, 6, r]
    assert n_ok > 20 and n_err > 5


SYNTHETIC_SEQ = """RDM;LEN "synthetic: every command the reader knows"
TITLE 'synthetic'
EPD   8.0
DIM   M
WL    656.3 587.6 486.1
REF   2
WTW   1 1 1
XAN   0.0 0.0
YAN   0.0 3.0
PRV
  PWL 700.0 600.0 500.0 400.0
  'MYGLASS' 1.6010 1.6060 1.6150 1.6330
END
SO    0.0 0.1e14
S     40.0 4.0 'MYGLASS'
  CIR 9.0
  CIR OBS 1.5
  ADX 0.25; ADY -0.5
S     -60.0 2.0 517.642
  ASP
  K   -0.7
  A   1.5e-06; B -2.0e-09; C 0.0; D&
   1.0e-14
  REX 8.0; REY 7.0
S     0.0 12.0
  STO
  XDE 0.1; YDE -0.2; ZDE 0.05
  ADE 2.0; BDE -1.5; CDE 0.5
S     0.0 -10.0 REFL
  CON
  K -1.0
  DAR
  ADE 12.0
  ELX 6.0; ELY 5.0
S     0.0 -5.0
  YTO
  BEN
  BDE 3.0
S     25.0 -3.0 NBK7_SCHOTT
  CUY 0.03
  DIF DOE
  HOR 1.0
  HWL 587.6; HCT R
  HCO C1 -0.002; HCO C3 1.0e-07
S     0.0 -20.0
  RDY -80.0
  REV
  XDE 0.3
  THI HMY 0.0
SI    0.0 0.0
GO
"""


@pytest.mark.needs_reference
def test_synthetic_seq_with_every_command(tmp_path):
    """one synthetic CODE V sequence with every command read_seq knows -- private catalogue,
    fictitious glass code, continuation lines, aspheric terms, all aperture and decenter
    commands (DAR / BEN / REV), a toroid, a radial DOE, solves -- through both importers"""
    from oracle import refshim
    refshim.install()
    from rayoptics.codev import cmdproc
    from rayoptics.seq.sequential import SequentialModel
    from rayoptics.optical.opticalmodel import OpticalModel
    from rayoptics_amd import SurfaceTable, ingest, abi
    path = tmp_path / 'synthetic.seq'
    path.write_text(SYNTHETIC_SEQ)
    saved = SequentialModel.set_clear_apertures, OpticalModel.update_model
    SequentialModel.set_clear_apertures = lambda self, **kw: None
    OpticalModel.update_model = lambda self, **kw: None
    try:
        opm, _info = cmdproc.read_lens(path, do_update=False)
    finally:
        SequentialModel.set_clear_apertures, OpticalModel.update_model = saved
    sm = opm['seq_model']
    sm.update_model()
    theirs = SurfaceTable.from_seq_model(sm)
    pres = ingest.read_seq(str(path))
    ours = pres.to_table(index_of=ingest.reference_fallback_index)
    assert ours.n_ifcs == theirs.n_ifcs == 9
    for t in (ours, theirs):
        t.n_table = t.n_table.copy()
        for r in t.rows:
            r.max_aperture = 1.0
    # private and fictitious glasses: dispersion is opticalglass's (not here); the named
    # catalogue glass is unknown on both sides (1.5)
    for i, m in enumerate(pres.media):
        if m[0] == 'model' or (m[0] == 'glass' and m[1] in pres.private_glasses):
            theirs.n_table[:, i] = ours.n_table[:, i]
    for i, m in enumerate(pres.media):
        if m[0] == 'mirror':
            theirs.n_table[:, i] = ours.n_table[:, i] = ours.n_table[:, i - 1]
    rows_equal(ours, theirs, 'synthetic.seq')
    assert ours.rows[5].profile == abi.PROFILE_NAMES['YToroid'] and ours.rows[6].ph.kind == abi.PH_DOE_RADIAL
    assert ours.rows[1].n_ap == 1 and ours.rows[2].n_ap == 1 and ours.rows[4].ap[0].kind == abi.AP_ALWAYS_BLOCK


def test_roa_decenter_records_ingested():
    """a .roa whose surfaces carry DecenterData ('dec and return' on a lens surface, a
    'decenter' coordinate break) -- tests/golden/decentered.roa, written in json_tricks'
    layout from a live reference model by make_golden.dump_roa -- gives, field by field, the
    table the reference's own transforms gave for that model (stored beside it)"""
    import json
    from rayoptics_amd import SurfaceTable, ingest
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    theirs = SurfaceTable.from_dict(json.load(open(os.path.join(here, 'decentered_roa_table.json'))))
    pres = ingest.read_roa(os.path.join(here, 'decentered.roa'))
    assert sum(s.decenter is not None for s in pres.ifcs) == 2
    ours = pres.to_table(wvls=theirs.wvls, index_of=ingest.reference_fallback_index)
    ours.n_table[:] = theirs.n_table            # (ModelGlass dispersion is the reference's)
    rows_equal(ours, theirs, 'decentered.roa')
    # the rotations are real ones
    assert any(abs(r.rt[1]) > 1e-3 for r in ours.rows)


@pytest.mark.needs_reference
def test_roa_decenter_fixture_reads_back_through_the_reference_classes():
    """the same file through the reference's own Surface / DecenterData / transform code
    (refmodels.load_roa) gives that table too: the fixture is a faithful .roa"""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import refmodels as rm
    from rayoptics_amd import SurfaceTable
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    stored = SurfaceTable.from_dict(json.load(open(os.path.join(here, 'decentered_roa_table.json'))))
    back = SurfaceTable.from_seq_model(rm.load_roa(os.path.join(here, 'decentered.roa'))['seq_model'])
    live = SurfaceTable.from_seq_model(rm.tilted_singlet()['seq_model'])
    for t in (stored, back, live):
        t.rows[0].max_aperture = t.rows[-1].max_aperture = 1.0
    rows_equal(back, stored, 'read back')
    rows_equal(live, stored, 'live model')

"""HIP engine + drop-in layer + LIVE reference in ONE process (GPU box).

The reference is pure Python; ``oracle/stage_reference.py`` builds it into ``oracle/_ref/`` as
sourceless byte code (git-ignored, travels with the gpurun snapshot like the built ``.so``
files), so on the GPU box -- where ``/root/reference`` does not exist -- the reference's own
``OpticalModel`` / figures / per-ray Python loop run next to the real HIP engine.  This closes
the transitive chain of DESIGN section 5: every test below compares what the reference's
consumers get with ``rayoptics_amd.install`` active (launches served by ``libroxtrace.so`` on
``cuda:0``) against what the *un-installed* reference computes in the same process.

Every test of tests/test_dropin_reference.py (the host-logic suite the build container runs
over the oracle-backed test double) is re-collected here over the HIP engine; the tests at the
bottom add larger direct comparisons.  Skipped only where ``oracle/_ref`` is absent.
"""
import numpy as np
import pytest

from oracle import refshim

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not refshim.available(),
                                 reason='reference not staged (oracle/_ref absent: run '
                                        'oracle/stage_reference.py where /root/reference exists)')]

import test_dropin_reference as _base            # noqa: E402
from test_dropin_reference import ref, both      # noqa: E402,F401  (fixture + helper)


@pytest.fixture()
def installed(ref):
    """the drop-ins over the product engine: engine factory None = engine.TraceEngine = HIP"""
    from rayoptics_amd import session, install
    from rayoptics_amd.engine import TraceEngine, load_library
    load_library()
    session._set_engine_factory(None)
    assert session._factory() is TraceEngine
    install.install()
    yield install
    install.uninstall()


# the whole host-logic suite, now over the HIP engine
for _name in dir(_base):
    if _name.startswith('test_'):
        globals()[_name] = getattr(_base, _name)
del _name


MODELS = ['dblgauss', 'nikkor', 'rc_telescope', 'zmx_evenasph_c3', 'cell_phone', 'tilted_singlet']


@pytest.mark.parametrize('model', MODELS)
def test_trace_grid_packets_hip_vs_the_references_python_loop(ref, installed, model):
    """`trace.trace_grid` (rayoptics/raytr/trace.py:563-605) with the identity filter: a 48 x 48
    pupil grid of the outermost field at every wavelength -- full packets, op_delta, the
    wavelength, the (vignetted) pupil coordinate, the None pattern of failed rays -- HIP engine
    vs the reference's per-ray loop, bit for bit, in one process"""
    import rayoptics.raytr.trace as trace
    opm = getattr(ref, model)()
    osp = opm['osp']
    fld = osp['fov'].fields[-1]
    n_ok = 0
    for wvl in osp['wvls'].wavelengths:
        def run():
            got = []        # (the filter keeps the packets: np.array() of ragged packets raises
            #                 in the reference under NumPy >= 1.24, trace.py:605)
            trace.trace_grid(opm, [np.array([-1., -1.]), np.array([1., 1.]), 48], fld, wvl, 0.0,
                             img_filter=lambda p, pkg: got.append((np.array(p), pkg)),
                             form='list', append_if_none=True)
            return got
        ours, theirs = both(installed, run)
        assert len(ours) == len(theirs) == 48 * 48
        for (po, ko), (pt, kt) in zip(ours, theirs):
            np.testing.assert_array_equal(po, pt)
            assert (ko is None) == (kt is None)
            if ko is None:
                continue
            n_ok += 1
            _base.same_pkg(ko, kt)
    assert n_ok > 300


@pytest.mark.parametrize('model', ['dblgauss', 'nikkor', 'rc_telescope', 'zmx_evenasph_c3'])
def test_spot_diagram_data_at_figure_size_hip_vs_reference(ref, installed, model):
    """SpotDiagramFigure at a size a user would draw (num_rays=32: 1 024 rays per field and
    wavelength through the fused packed-hits launch) == the reference's figure data"""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    from rayoptics.mpl.axisarrayfigure import SpotDiagramFigure
    opm = getattr(ref, model)()

    def run():
        fig = plt.figure(FigureClass=SpotDiagramFigure, opt_model=opm, num_rays=32)
        fig.update_data()
        data = [[np.array(g) for g in row[0][0]] for row in fig.axis_data_array]
        plt.close(fig)
        return data
    ours, theirs = both(installed, run)
    n = 0
    for ro, rt_ in zip(ours, theirs):
        assert len(ro) == len(rt_)
        for go, gt in zip(ro, rt_):
            np.testing.assert_array_equal(go, gt)
            n += len(go)
    assert n > 1000


def test_the_engine_behind_the_drop_ins_is_the_hip_library(ref, installed):
    """what served the launches above: a TraceEngine on cuda:0 holding a handle of
    libroxtrace.so -- not the oracle-backed test double, not a CPU path"""
    from rayoptics_amd import session
    from rayoptics_amd.engine import TraceEngine
    import rayoptics.raytr.trace as trace
    opm = ref.dblgauss()
    fld = opm['osp']['fov'].fields[1]
    trace.trace_grid(opm, [np.array([-1., -1.]), np.array([1., 1.]), 5], fld,
                     opm['seq_model'].central_wavelength(), 0.0, form='list',
                     img_filter=lambda p, pkg: [p[0], p[1], np.nan if pkg is None else pkg[1]])
    eng = session.engine_for(opm)
    assert type(eng) is TraceEngine and str(eng.device).startswith('cuda')
    with open('/proc/self/maps') as f:
        maps = f.read()
    assert 'libroxtrace.so' in maps


# ---------------------------------------------------------------- tolerance mode through the drop-ins
@pytest.mark.parametrize('model', ['dblgauss', 'nikkor', 'rc_telescope', 'zmx_evenasph_c3', 'cell_phone'])
def test_tolerance_mode_figures_within_1e_10_of_the_reference(ref, installed, model):
    """rayoptics_amd.install(tolerance_mode=True): the reference's own SpotDiagramFigure (packed
    hits), Wavefront grid (OPD epilogue) and RayFanFigure data come from the ROX_FAST_FP64
    kernels -- every number within 1e-10 * max(1, |reference|) of the un-installed reference's,
    the survivor counts identical (north_star's tolerance; the default path is bit-exact)"""
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    from rayoptics.mpl.axisarrayfigure import SpotDiagramFigure, RayFanFigure
    from rayoptics_amd import session
    opm = getattr(ref, model)()

    def run():
        out = []
        fig = plt.figure(FigureClass=SpotDiagramFigure, opt_model=opm, num_rays=24)
        fig.update_data()
        out += [np.array(g, dtype=float) for row in fig.axis_data_array for g in row[0][0]]
        plt.close(fig)
        for dt in ('Ray', 'OPD'):
            fig = plt.figure(FigureClass=RayFanFigure, opt_model=opm, data_type=dt, num_rays=21)
            fig.update_data()
            for row in fig.axis_data_array:
                for fan in row:
                    for curve in fan[0]:
                        out.append(np.array(curve[0], dtype=float))
                        out.append(np.array(curve[1], dtype=float))
            plt.close(fig)
        return out
    was = session.set_tolerance_mode(True)
    try:
        ours, theirs = both(installed, run)
    finally:
        session.set_tolerance_mode(was)
    assert len(ours) == len(theirs) and len(ours) > 6
    worst, n = 0.0, 0
    for a, b in zip(ours, theirs):
        assert a.shape == b.shape
        m = np.isfinite(b)
        assert np.array_equal(m, np.isfinite(a))
        if m.any():
            worst = max(worst, float((np.abs(a[m] - b[m]) / np.maximum(1.0, np.abs(b[m]))).max()))
            n += int(m.sum())
    assert n > 1000 and worst <= 1e-10, (model, worst)
    import helpers as H
    H.record('tolerance_mode_figures_vs_live_reference', model=model, values=n, worst_scaled_error=worst)

"""TEST INFRASTRUCTURE ONLY -- not part of the product path.

Makes the *reference* (mjhoptics/ray-optics, pure Python, mounted read-only at
/root/reference) importable in the build container, where six of its
third-party dependencies (``opticalglass``, ``anytree``, ``transforms3d``,
``json_tricks``, ``parsimonious``, ``deprecation``) are not installed.

Nothing here re-implements reference arithmetic: the stubs only satisfy
``import`` statements, plus a constant-index ``opticalglass.opticalmedium`` /
``modelglass`` so that ``SequentialModel`` can hold media (both sides of every
parity test read the *same* evaluated ``seq_model.rndx`` table, so the glass
dispersion model is irrelevant to trace parity -- SURVEY.md section 8c).

The one piece of third-party arithmetic that *is* restated is
``transforms3d.euler.euler2mat(..., 'rxyz')`` (used by
``rayoptics/util/misc_math.py:150-160`` for tilted surfaces); it is a plain
product of three axis rotations and is flagged "tilt parity unpinned" in
DESIGN.md.

``/root/reference`` does not exist on the GPU box.  There the reference is
importable only from ``oracle/_ref`` -- sourceless byte code built by
``oracle/stage_reference.py`` (the Python analogue of a compiled reference
``.so``; git-ignored, travels with the gpurun snapshot).  Callers of
:func:`install`: the golden-vector generators, the ``needs_reference`` CPU
tests, the live-reference ``-m gpu`` tests and ``bench.py``'s ``cpu_baseline``.
"""
import os
import sys
import types
import math

STAGED = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref')


def _staged_ok() -> bool:
    """oracle/_ref holds the reference as sourceless byte code (oracle/stage_reference.py);
    usable only by the interpreter version that compiled it"""
    import importlib.util
    import json
    try:
        with open(os.path.join(STAGED, 'stamp.json')) as f:
            return json.load(f).get('magic') == importlib.util.MAGIC_NUMBER.hex()
    except (OSError, ValueError):
        return False


def _resolve():
    env = os.environ.get('ROX_REFERENCE_SRC')
    if env == 'staged':
        return STAGED
    if env:
        return env
    if os.path.isdir('/root/reference/src/rayoptics'):
        return '/root/reference/src'
    return STAGED if _staged_ok() else '/root/reference/src'


REFERENCE_SRC = _resolve()


def available() -> bool:
    d = os.path.join(REFERENCE_SRC, 'rayoptics')
    return os.path.isfile(os.path.join(d, '__init__.py')) or (
        os.path.isfile(os.path.join(d, '__init__.pyc')) and _staged_ok())


def is_staged() -> bool:
    return os.path.abspath(REFERENCE_SRC) == STAGED


class _Permissive(types.ModuleType):
    """module whose unknown attributes are fresh dummy classes"""
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        cls = type(name, (object,), {'__init__': lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _stub(name):
    if name in sys.modules:
        return sys.modules[name]
    m = _Permissive(name)
    sys.modules[name] = m
    if '.' in name:
        parent, child = name.rsplit('.', 1)
        setattr(_stub(parent), child, m)
    return m


# --- constant-index media (import plumbing; no dispersion model) -----------
class OpticalMedium:
    def __init__(self, nd=1.0, lbl='', cat=''):
        self.label = lbl
        self.n = nd
        self._catalog_name = cat

    def name(self):
        return self.label

    def catalog_name(self):
        return self._catalog_name

    def rindex(self, wv_nm):
        return self.n

    def calc_rindex(self, wv_nm):
        return self.n

    def meas_rindex(self, wvl):
        return self.n

    def sync_to_restore(self):
        pass


class Air(OpticalMedium):
    def __init__(self):
        super().__init__(1.0, 'air', '')

    def __repr__(self):
        return 'Air()'


class ConstantIndex(OpticalMedium):
    def __init__(self, nd, lbl, cat=''):
        super().__init__(nd, lbl, cat)


class InterpolatedMedium(OpticalMedium):
    """a glass given by index samples (CODE V private catalogue, cmdproc.py:146-158).
    opticalglass interpolates them with its own scheme; this stand-in is linear in
    wavelength -- parity tests neutralise the indices behind such glasses."""

    def __init__(self, label='', pairs=None, rndx=None, wvls=None, cat=''):
        super().__init__(1.5, str(label), cat)
        if pairs is not None:
            wvls = [p[0] for p in pairs]
            rndx = [p[1] for p in pairs]
        self.wvls = list(wvls) if wvls is not None else []
        self.rndx = list(rndx) if rndx is not None else []

    def rindex(self, wv_nm):
        import numpy as np
        if not self.wvls:
            return self.n
        w = get_wavelength(wv_nm)
        order = np.argsort(self.wvls)
        return float(np.interp(w, np.asarray(self.wvls)[order], np.asarray(self.rndx)[order]))

    calc_rindex = meas_rindex = rindex


class ModelGlass(OpticalMedium):
    """synthetic test glass: linear dispersion n(w) = nd + (nd-1)/vd *
    (lam_d - w)/(lam_C - lam_F), so that nF - nC = (nd-1)/vd.  NOT
    opticalglass's Buchdahl model -- it only has to give the wavelengths
    different indices; both sides of a parity test read the same rndx."""

    def __init__(self, nd, vd, mat, cat='user'):
        super().__init__(nd, str(mat), cat)
        self.vd = vd

    def rindex(self, wv_nm):
        if not self.vd:
            return self.n
        w = get_wavelength(wv_nm)
        return self.n + (self.n - 1.0) / self.vd * (587.5618 - w) / (656.2725 - 486.1327)


_SPECTRAL = {'F': 486.1327, 'd': 587.5618, 'C': 656.2725, 'e': 546.074,
             "F'": 479.9914, "C'": 643.8469, 'g': 435.8343, 'h': 404.6561,
             'r': 706.5188, 'D': 589.2938, 's': 852.11, 't': 1013.98,
             'i': 365.014, 'He-Ne': 632.8}


def get_wavelength(wvl):
    if isinstance(wvl, (int, float)):
        return float(wvl)
    return _SPECTRAL[wvl]


def _euler2mat(ai, aj, ak, axes='sxyz'):
    """restatement of transforms3d.euler.euler2mat for axes='rxyz' only
    (intrinsic x, then y, then z):  R = Rx(ai) . Ry(aj) . Rz(ak)."""
    import numpy as np
    assert axes == 'rxyz'
    cx, sx = math.cos(ai), math.sin(ai)
    cy, sy = math.cos(aj), math.sin(aj)
    cz, sz = math.cos(ak), math.sin(ak)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=float)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=float)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=float)
    return rx @ ry @ rz


class _Node:
    """import-plumbing stand-in for anytree.Node (name, parent, children,
    arbitrary attributes); the part tree is never consulted on the trace
    path."""

    def __init__(self, name='', parent=None, children=None, **kwargs):
        self.name = name
        self.parent = parent
        self.children = list(children) if children else []
        for k, v in kwargs.items():
            setattr(self, k, v)
        if parent is not None and hasattr(parent, 'children'):
            parent.children.append(self)


class _Counter(dict):
    """opticalglass.util.Counter: a dict whose missing keys count from zero"""

    def __missing__(self, key):
        return 0


_installed = False


def install():
    """put the reference on sys.path with its missing dependencies stubbed."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f'reference not found under {REFERENCE_SRC}')

    for name in ('anytree', 'anytree.exporter', 'anytree.importer',
                 'anytree.search', 'json_tricks', 'transforms3d',
                 'transforms3d.euler', 'deprecation', 'parsimonious',
                 'parsimonious.grammar', 'parsimonious.nodes', 'ipywidgets',
                 'opticalglass', 'opticalglass.glassfactory',
                 'opticalglass.glasserror', 'opticalglass.glass',
                 'opticalglass.rindexinfo', 'opticalglass.glasspolygons',
                 'opticalglass.util', 'opticalglass.glassmap',
                 'opticalglass.glassmapviewer', 'opticalglass.opticalmedium',
                 'opticalglass.modelglass', 'opticalglass.spectral_lines'):
        _stub(name)

    sys.modules['anytree'].Node = _Node
    sys.modules['transforms3d.euler'].euler2mat = _euler2mat
    # `@deprecation.deprecated(...)` decorator must be transparent
    sys.modules['deprecation'].deprecated = \
        lambda *a, **k: (lambda f: f)

    om = sys.modules['opticalglass.opticalmedium']
    om.OpticalMedium = OpticalMedium
    om.Air = Air
    om.ConstantIndex = ConstantIndex
    om.InterpolatedMedium = InterpolatedMedium
    sys.modules['opticalglass.modelglass'].ModelGlass = ModelGlass
    sys.modules['opticalglass.spectral_lines'].get_wavelength = get_wavelength
    ge = sys.modules['opticalglass.glasserror']
    for exc in ('GlassError', 'GlassNotFoundError', 'GlassCatalogNotFoundError'):
        setattr(ge, exc, type(exc, (Exception,), {}))
    # an empty glass catalogue: every named glass is "not found", which the
    # reference's importers turn into ConstantIndex(1.5, 'not NAME')
    # (rayoptics/seq/medium.py:172-203); geometry import is unaffected
    gf = sys.modules['opticalglass.glassfactory']
    gf._cat_names = []
    gf._cat_names_uc = []
    gf._custom_glass_registry = {}

    def _create_glass(name, catalog):
        raise ge.GlassNotFoundError(name)

    def _get_glass_catalog(name):
        raise ge.GlassCatalogNotFoundError(name)
    gf.create_glass = _create_glass
    gf.get_glass_catalog = _get_glass_catalog
    gf.register_glass = lambda mat: None
    sys.modules['opticalglass.util'].Counter = _Counter
    gl = sys.modules['opticalglass.glass']
    gl.Robb1983Catalog = lambda: type('C', (), {'glass_list': []})()
    gl.decode_glass_name = lambda name: ((name[:1] or '?', name[1:]), '', '')

    _redirect_importer_logs()
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    _installed = True


def _redirect_importer_logs():
    """The reference's importers open ``logging.FileHandler('zmx_read_lens.log')`` /
    ``('cv_cmd_proc.log')`` at import time (rayoptics/zemax/zmxread.py:31-33,
    rayoptics/codev/cmdproc.py:30-32): a relative name, resolved against the
    current directory -- i.e. into this repository whenever a test imports them.
    Relative ``*.log`` handlers are sent to a scratch directory instead."""
    import logging
    import tempfile
    if getattr(logging.FileHandler, '_rox_redirected', False):
        return
    scratch = tempfile.mkdtemp(prefix='rox_ref_logs_')
    orig = logging.FileHandler.__init__

    def init(self, filename, *a, **k):
        name = os.fspath(filename)
        if not os.path.isabs(name) and name.endswith('.log'):
            filename = os.path.join(scratch, os.path.basename(name))
        orig(self, filename, *a, **k)
    logging.FileHandler.__init__ = init
    logging.FileHandler._rox_redirected = True

"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads
and exports every symbol include/roxtrace.h declares; struct layouts agree.
No compute is launched (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

from rayoptics_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'rox_build', os.path.join(ROOT, 'ray-optics_amd', 'build.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    path = b.build()
    return C.CDLL(path)


def header_symbols(header='roxtrace.h'):
    with open(os.path.join(ROOT, 'include', header)) as f:
        src = f.read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(rox_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(abi.EXPORTS)
    assert header_symbols('roxtrace_diag.h') == sorted(abi.DIAG_EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in header_symbols() + header_symbols('roxtrace_diag.h'):
        assert hasattr(lib, name), f'{name} not exported'


def test_dynamic_symbol_table_is_the_c_abi_and_nothing_else(lib):
    """`nm -D` of libroxtrace.so == the functions the two headers declare: no rox:: internals,
    kernel host stubs or template instantiations (-fvisibility=hidden + csrc/libroxtrace.map),
    and DT_SONAME carries the ABI version"""
    import subprocess
    path = lib._name
    out = subprocess.check_output(['nm', '-D', '--defined-only', path], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == sorted(header_symbols() + header_symbols('roxtrace_diag.h'))
    dyn = subprocess.check_output(['readelf', '-d', path], text=True)
    assert f'[libroxtrace.so.{abi.ABI_VERSION}]' in dyn
    link = os.path.join(os.path.dirname(path), f'libroxtrace.so.{abi.ABI_VERSION}')
    assert os.path.exists(link)


def test_abi_version_and_error_string(lib):
    abi.declare(lib)
    assert lib.rox_abi_version() == abi.ABI_VERSION
    # argument errors are reported without touching a device
    rc = lib.rox_system_create(None, 0, None, None, 0, None)
    assert rc == -1
    assert b'rox_system_create' in lib.rox_last_error()


def test_engine_fails_loudly_without_gpu():
    import torch
    from rayoptics_amd import SurfaceTable
    from rayoptics_amd.engine import TraceEngine, EngineError
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    tbl = SurfaceTable.from_prescription(
        [dict(cv=0, thi=10.0), dict(cv=0.02, thi=3.0, n=1.5), dict(cv=0, thi=0)])
    with pytest.raises(EngineError):
        TraceEngine(tbl)


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of every struct as a C compiler sees include/roxtrace.h
    against the ctypes mirror in abi.py"""
    import subprocess
    structs = {'rox_aperture': abi.Aperture, 'rox_phase': abi.Phase, 'rox_surface': abi.Surface,
               'rox_wavefront': abi.Wavefront, 'rox_opts': abi.Opts, 'rox_field': abi.Field,
               'rox_grid': abi.Grid, 'rox_out': abi.Out, 'rox_aim': abi.Aim, 'rox_vig': abi.Vig,
               'rox_enp': abi.Enp}
    lines = ['#include <stdio.h>', '#include <stddef.h>',
             f'#include "{ROOT}/include/roxtrace.h"', 'int main(void) {']
    for cname, st in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _t in st._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['return 0; }']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-o', str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, st in structs.items():
        assert int(got[cname]) == C.sizeof(st), cname
        for fname, _t in st._fields_:
            assert int(got[f'{cname}.{fname}']) == getattr(st, fname).offset, f'{cname}.{fname}'


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: no module of the product package (nor the examples)
    imports, loads or names it; importing the package does not pull it in either"""
    import ast
    import subprocess
    import sys
    pkg = os.path.join(ROOT, 'ray-optics_amd')
    files = [os.path.join(pkg, f) for f in os.listdir(pkg) if f.endswith('.py')]
    files += [os.path.join(ROOT, 'rayoptics_amd.py')]
    files += [os.path.join(ROOT, 'examples', f) for f in os.listdir(os.path.join(ROOT, 'examples'))
              if f.endswith('.py')]
    for path in files:
        tree = ast.parse(open(path).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ''] + [a.name for a in node.names]
            assert not any(n.split('.')[0] == 'oracle' or 'rox_oracle' in n for n in names), path
            if isinstance(node, ast.Constant) and isinstance(node.value, str):
                assert 'librox_oracle' not in node.value, path
    for src in os.listdir(os.path.join(pkg, 'csrc')):
        assert 'oracle' not in open(os.path.join(pkg, 'csrc', src)).read().lower(), src
    code = ('import sys; sys.path.insert(0, %r); import rayoptics_amd; '
            'import rayoptics_amd.trace, rayoptics_amd.analyses, rayoptics_amd.ingest, '
            'rayoptics_amd.dist, rayoptics_amd.vigcalc; '
            'assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules), '
            '[m for m in sys.modules if m.startswith("oracle")]' % ROOT)
    subprocess.check_call([sys.executable, '-c', code])


def test_bench_scaling_keys_are_what_they_say():
    """bench.py's self-description of a `--gpus N` line: speedup = one_gpu_same_problem_ms /
    ms_per_step and efficiency = speedup / n_gpus (no GPU needed: the helper is arithmetic)"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('rox_bench', os.path.join(root, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    k = b.scaling_keys(77.0, 11.0, 8)
    assert k['one_gpu_same_problem_ms'] == 77.0
    assert k['speedup'] == 77.0 / 11.0 and k['efficiency'] == k['speedup'] / 8
    k1 = b.scaling_keys(77.0, 77.0, 1)
    assert k1['speedup'] == 1.0 and k1['efficiency'] == 1.0
    assert b.scaling_keys(None, 11.0, 4)['speedup'] is None


def test_spot_stats_rejects_bad_arguments_without_a_device(lib):
    """rox_spot_stats validates before it touches the device: null buffers, an unknown layout,
    a histogram without edges, edges that do not increase"""
    import numpy as np
    abi.declare(lib)
    summ = abi.SpotSummary()
    xy = np.zeros(8)
    hist = np.zeros(4, dtype=np.uint32)
    good = np.array([0.0, 1.0, 2.0])
    bad = np.array([0.0, 2.0, 1.0])
    assert lib.rox_spot_stats(None, 4, None, None, 4, abi.SPOT_ROWS, None, 0, None, 0, C.byref(summ), None, None) == -1
    assert lib.rox_spot_stats(xy.ctypes.data, 4, None, None, 4, 7, None, 0, None, 0, C.byref(summ), None, None) == -1
    assert lib.rox_spot_stats(xy.ctypes.data, 2, None, None, 4, abi.SPOT_ROWS, None, 0, None, 0, C.byref(summ), None,
                              None) == -1          # ld < n
    assert lib.rox_spot_stats(xy.ctypes.data, 4, None, None, 4, abi.SPOT_ROWS, None, 0, None, 0, C.byref(summ),
                              hist.ctypes.data, None) == -1     # histogram without edges
    assert lib.rox_spot_stats(xy.ctypes.data, 4, None, None, 4, abi.SPOT_ROWS, bad.ctypes.data, 3, good.ctypes.data, 3,
                              C.byref(summ), hist.ctypes.data, None) == -1
    assert b'monotonically' in lib.rox_last_error()


def test_tolerance_mode_is_an_opt_in_of_the_drop_in_layer():
    """session.set_tolerance_mode / install(tolerance_mode=...): ROX_FAST_FP64 rides on every rox_opts
    the drop-ins build, and only then"""
    from rayoptics_amd import session, trace
    assert session.TOLERANCE_MODE is False
    o = trace.opts_from_kwargs(13, {'check_apertures': True}, abi.OUT_HITS)
    assert not (o.flags & abi.FAST_FP64) and (o.flags & abi.CHECK_APERTURES)
    was = session.set_tolerance_mode(True)
    try:
        assert was is False
        o = trace.opts_from_kwargs(13, {}, abi.OUT_HITS)
        assert o.flags & abi.FAST_FP64
    finally:
        session.set_tolerance_mode(was)
    assert not (trace.opts_from_kwargs(13, {}, abi.OUT_FULL).flags & abi.FAST_FP64)

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


def header_symbols():
    with open(os.path.join(ROOT, 'include', 'roxtrace.h')) as f:
        src = f.read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(rox_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(abi.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in header_symbols():
        assert hasattr(lib, name), f'{name} not exported'


def test_abi_version_and_error_string(lib):
    abi.declare(lib)
    assert lib.rox_abi_version() == abi.ABI_VERSION
    # argument errors are reported without touching a device
    rc = lib.rox_system_create(None, 0, None, 0, None)
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


def test_struct_sizes_match_header():
    # sizes asserted in abi.py against the C header's static layout
    assert C.sizeof(abi.Surface) == 408 and C.sizeof(abi.Aperture) == 40
    assert C.sizeof(abi.Opts) == 352 and C.sizeof(abi.Field) == 96
    assert C.sizeof(abi.Grid) == 48 and C.sizeof(abi.Out) == 48

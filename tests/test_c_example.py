"""examples/spot_diagram.c: the C ABI used from plain C (gcc, include/roxtrace.h, -lroxtrace),
no Python, torch or HIP headers on the caller's side.

CPU: the file compiles and links against the in-tree library, and without a GPU the program
ends with the library's own error.  GPU: the pairs it dumps equal the CPU oracle's for the same
table -- typed in a second time here through the Python structures -- bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import rayoptics_amd  # noqa: F401
from rayoptics_amd import abi, SurfaceTable
from rayoptics_amd.engine import make_opts, make_grid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'ray-optics_amd')

CV = [0.0, 1.0 / 50.0, -1.0 / 50.0, 0.0]
THI = [100.0, 5.0, 95.0, 0.0]
NDX = [1.0, 1.5168, 1.0, 1.0]
SEMI_AP = [1.0e10, 9.0, 9.0, 1.0e10]


@pytest.fixture(scope='module')
def exe(tmp_path_factory):
    from rayoptics_amd import build
    build.build()
    out = tmp_path_factory.mktemp('c_example') / 'spot_diagram'
    subprocess.check_call(['gcc', '-std=c99', '-O2', '-Wall', '-Wextra', '-Werror',
                           '-I' + os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'spot_diagram.c'),
                           '-L' + PKG, '-lroxtrace', '-Wl,-rpath,' + PKG, '-lm', '-o', str(out)])
    return str(out)


def the_table():
    rows = (abi.Surface * 4)()
    for i, row in enumerate(rows):
        row.mode = abi.DUMMY if i in (0, 3) else abi.TRANSMIT
        row.profile = abi.SPHERICAL
        row.rt_order = abi.RT_C_ORDER
        row.cv, row.ec = CV[i], 1.0
        row.rt[0] = row.rt[4] = row.rt[8] = 1.0
        row.t[2] = THI[i]
        row.z_dir = 1.0
        row.max_aperture = SEMI_AP[i]
    return SurfaceTable(rows, np.array([NDX]), [587.5618])


def test_c_example_compiles_and_fails_loudly_without_a_gpu(exe):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is here: test_c_example_equals_the_oracle runs it')
    r = subprocess.run([exe, '16'], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert 'rox_device_count' in r.stderr and 'no ROCm-capable device' in r.stderr


@pytest.mark.gpu
def test_c_example_equals_the_oracle(exe, tmp_path):
    from oracle import oracle
    num = 96
    dump = tmp_path / 'pairs.bin'
    r = subprocess.run([exe, str(num), str(dump)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count('rays reach the image') == 2
    raw = dump.read_bytes()
    tbl = the_table()
    off = 0
    for y in (0.0, 5.0):
        n = int(np.frombuffer(raw, dtype=np.int64, count=1, offset=off)[0])
        got = np.frombuffer(raw, dtype=np.float64, count=2 * n, offset=off + 8).reshape(n, 2)
        off += 8 + 16 * n
        fld = abi.Field()
        fld.kind = abi.FLD_EPD
        fld.pt0[1] = y
        fld.eprad, fld.z_enp, fld.z_dir0 = 8.0, THI[0], 1.0
        opts = make_opts(flags=abi.CHECK_APERTURES | abi.INTERSECT_OBJ | abi.APPLY_VIGNETTING,
                         out_mode=abi.OUT_HITS, first_surf=1, last_surf=2)
        res = oracle.trace_pupil_grid(tbl, fld, make_grid((-1., -1.), (1., 1.), num), 0, opts)
        ok = res.status == abi.OK
        want = res.seg[:, ok].T
        assert 0 < n < num * num and n == int(ok.sum())
        assert np.array_equal(got.view(np.int64), np.ascontiguousarray(want).view(np.int64))
    assert off == len(raw)

"""-m gpu: the ctypes stub INTEGRATION.md shows a ray-optics maintainer is executed as
written (its code block is lifted out of the document) against the built library: struct
layouts, argtypes and the host-pointer call in it are real, and its result equals the oracle's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rayoptics_amd import abi
import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_documented_ctypes_stub_runs():
    from oracle import oracle
    from rayoptics_amd.engine import load_library, LIB_PATH
    load_library()                                      # builds libroxtrace.so if need be
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = re.search(r'```python\n(.*?)```', text, flags=re.S).group(1)
    assert "C.CDLL('libroxtrace.so')" in block
    ns = {}
    exec(block.replace("C.CDLL('libroxtrace.so')", f'C.CDLL({LIB_PATH!r})'), ns)
    # the stub's structs have the header's sizes
    assert C.sizeof(ns['Surface']) == C.sizeof(abi.Surface) == 576
    assert C.sizeof(ns['Opts']) == C.sizeof(abi.Opts) and C.sizeof(ns['Out']) == C.sizeof(abi.Out)
    fx = H.fixture('dblgauss')
    tbl = fx.table
    N = tbl.n_ifcs
    lib = ns['lib']
    handle = C.c_void_p()
    rows = (ns['Surface'] * N).from_buffer_copy(bytes(tbl.rows))
    wv = np.array(tbl.wvls, dtype=float)
    rc = lib.rox_system_create(rows, N, tbl.n_table.ctypes.data, wv.ctypes.data, len(tbl.wvls),
                               C.byref(handle))
    assert rc == 0, lib.rox_last_error()
    cr = fx['rays_ap']
    pt0 = np.ascontiguousarray(cr['pt0'])
    dir0 = np.ascontiguousarray(cr['dir0'])
    seg, op, status, fail = ns['trace_list_of_rays_soa'](handle, pt0, dir0, 1, N, check_apertures=True)
    opts = oracle.make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES, first_surf=1, last_surf=N - 2)
    orc = oracle.trace_rays(tbl, pt0, dir0, 1, opts)
    np.testing.assert_array_equal(status, orc.status)
    np.testing.assert_array_equal(fail, orc.fail_surf)
    same = (seg == orc.seg) | (np.isnan(seg) & np.isnan(orc.seg))
    assert same.all()
    assert np.array_equal(op, orc.op, equal_nan=True)
    lib.rox_system_destroy.argtypes = [C.c_void_p]
    lib.rox_system_destroy(handle)

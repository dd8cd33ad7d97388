"""shared test helpers: golden-fixture loading and SoA comparison"""
import json
import os

import numpy as np

from rayoptics_amd import SurfaceTable, abi
from rayoptics_amd.table import field_struct

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# north_star: "matching the reference NumPy path within 1e-10 on ray
# intercepts".  Intercepts and direction cosines of interior surfaces are
# O(1..100); object-space quantities of infinite-conjugate systems are O(1e10)
# (ulp 2e-6), so the bound is applied relative to max(1, |ref|).
ATOL = 1e-10


class Fixture:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.table = SurfaceTable.from_dict(json.loads(str(z['table_json'])))
        self.cases = {}
        for key in z.files:
            if '/' in key:
                c, k = key.split('/', 1)
                self.cases.setdefault(c, {})[k] = z[key]

    def __getitem__(self, case):
        return self.cases[case]


_cache = {}


def fixture(name):
    if name not in _cache:
        _cache[name] = Fixture(name)
    return _cache[name]


def field_from_arr(a):
    return field_struct(a[0:3], a[3:5], a[5], a[6], a[7:11], a[11])


def make_opts(case, out_mode=abi.OUT_FULL, **kw):
    from oracle.oracle import make_opts as mk
    return mk(flags=int(case['flags']) if 'flags' in case else abi.INTERSECT_OBJ,
              out_mode=out_mode,
              first_surf=int(case.get('first_surf', 0)),
              last_surf=int(case.get('last_surf', -1)), **kw)


def assert_soa_close(exp_seg, got_seg, what='seg', atol=ATOL, require_exact=False):
    """NaN pattern identical; values within atol*max(1,|ref|).  Returns the
    fraction of bit-identical finite entries."""
    exp_seg = np.asarray(exp_seg)
    got_seg = np.asarray(got_seg)
    assert exp_seg.shape == got_seg.shape, (what, exp_seg.shape, got_seg.shape)
    en, gn = np.isnan(exp_seg), np.isnan(got_seg)
    assert np.array_equal(en, gn), f'{what}: NaN pattern differs at {np.argwhere(en != gn)[:5]}'
    m = ~en
    if not m.any():
        return 1.0
    e, g = exp_seg[m], got_seg[m]
    err = np.abs(e - g) / np.maximum(1.0, np.abs(e))
    worst = err.max()
    assert worst <= atol, f'{what}: max scaled error {worst:.3e} > {atol}'
    exact = float(np.mean(e == g))
    if require_exact:
        assert exact == 1.0, f'{what}: only {exact:.6f} bit-identical'
    return exact


def assert_result_matches(case, res, atol=ATOL, require_exact=False, seg_key='seg'):
    """compare an oracle/device result object (seg, op, status, fail_surf) with
    the reference outputs stored in a golden case"""
    np.testing.assert_array_equal(res.status, case['status'])
    if 'fail_surf' in case:
        np.testing.assert_array_equal(res.fail_surf, case['fail_surf'])
    K = case[seg_key].shape[0]
    f1 = assert_soa_close(case[seg_key], np.asarray(res.seg)[:K], 'seg', atol, require_exact)
    f2 = assert_soa_close(case['op'], res.op, 'op', atol, require_exact)
    return min(f1, f2)

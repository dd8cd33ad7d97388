#!/usr/bin/env python3
"""Steady-state kernel time of the ROX_OUT_OPD / ROX_OUT_FAN-shaped launches (a num x num wavefront
grid with the OPD epilogue) on the golden wavefront cases, bit-exact and in tolerance mode.

    [ROX_LIB=variant.so] python tools/opd_probe.py [--num 1024]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--num', type=int, default=1024)
    args = ap.parse_args()
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, engine
    from rayoptics_amd.engine import TraceEngine, make_grid, DeviceResult
    import helpers as H
    from test_oracle_golden import opd_opts
    out = {'lib': os.path.basename(engine.LIB_PATH), 'num': args.num}
    for name, case in (('dblgauss', 'opd_f2'), ('nikkor', 'opd_f1'), ('telecentric', 'opd_f2')):
        fx = H.fixture(name)
        c = fx[case]
        eng = TraceEngine(fx.table)
        fld = H.field_from_arr(c['field'])
        grid = make_grid(c['start'], c['stop'], args.num)
        R = args.num ** 2
        res = DeviceResult(torch, eng.device, 0, R, abi.OUT_OPD, want_pupil=False, nan_fill=False)
        for tag, extra in (('exact', 0), ('tolerance', abi.FAST_FP64)):
            o = opd_opts(c)
            o.flags |= extra
            ms = eng.time_pupil_grid_sustained(fld, grid, int(c['wvl_idx']), o, res, 20)
            out[f'{name}/{case} {tag} us'] = round(ms * 1e3, 1)
        eng.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()

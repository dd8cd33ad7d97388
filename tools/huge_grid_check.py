#!/usr/bin/env python3
"""One pupil grid of 65536 x 65536 = 2^32 rays in ONE launch (ray indices far beyond int32, 73 GB
of HITS output on the 288 GB part): rays with indices above 2^31 equal the same rows traced as a
row block and the oracle's for the same pupil coordinates, bit for bit."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(num=65536):
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    from oracle import oracle
    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    fld = wl.fields[2]
    opts = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING, out_mode=abi.OUT_HITS,
                     first_surf=1, last_surf=N - 2, foc=wl.foc, image_pt=wl.image_pts[2])
    t0 = time.perf_counter()
    res = eng.trace_pupil_grid(fld, make_grid((-1., -1.), (1., 1.), num), 0, opts, want_pupil=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    R = num * num
    n_ok = int((res.status == abi.OK).sum().item())
    # four rows at 0.6 num (ray indices ~2.6e9 > 2^31), traced again as a row block (indices from 0)
    rows = 4
    r0 = int(0.6 * num)
    lo, hi = r0 * num, (r0 + rows) * num
    blk = eng.trace_pupil_grid(fld, make_grid((-1., -1.), (1., 1.), num, row_begin=r0, row_count=rows),
                               0, opts, want_pupil=True)
    a = res.seg[:, lo:hi].cpu().numpy()
    b = blk.seg[:, :rows * num].cpu().numpy()
    sa = res.status[lo:hi].cpu().numpy()
    sb = blk.status[:rows * num].cpu().numpy()
    same_block = bool(np.array_equal(sa, sb) and np.array_equal(a[:, sa == 0].view(np.int64), b[:, sb == 0].view(np.int64)))
    # ... and against the oracle through their pupil coordinates
    px = res.pupil[0, lo:hi].cpu().numpy()
    py = res.pupil[1, lo:hi].cpu().numpy()
    sel = np.r_[0:64, rows * num - 64:rows * num, np.arange(0, rows * num, 4099)]
    o2 = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES, out_mode=abi.OUT_HITS, first_surf=1, last_surf=N - 2,
                   foc=wl.foc, image_pt=wl.image_pts[2])     # (the coordinates are already vignetted)
    orc = oracle.trace_pupil_list(wl.table, fld, px[sel], py[sel], 0, o2)
    ok = orc.status == abi.OK
    same_oracle = bool(np.array_equal(orc.status, sa[sel]) and
                       np.array_equal(orc.seg[:, ok].view(np.int64), a[:, sel][:, ok].view(np.int64)))
    print(json.dumps({'num': num, 'rays': R, 'first_compared_ray_index': lo, 'above_int32': lo > 2 ** 31, 'launch_ms': ms,
                      'rays_through': n_ok, 'hits_GB': round((16 * R + R) / 1e9, 1),
                      'rows_equal_their_row_block': same_block, 'sampled_rays_equal_the_oracle': same_oracle,
                      'oracle_rays_compared': int(len(sel)), 'of_them_through': int(ok.sum())}))
    eng.close()


def full(num=8192):
    """FULL packets of an 8192 x 8192 grid (67 M rays, 72 GB: element offsets up to 8.7e9 in
    seg[13][10][ld]): the last four rows equal the same rows traced as a row block, every
    segment of every ray, and their hits equal the HITS launch's"""
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    wl = workloads.load('dblgauss_c2')
    N = wl.n_ifcs
    eng = TraceEngine(wl.table)
    fld = wl.fields[0]
    opts = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING, out_mode=abi.OUT_FULL,
                     first_surf=1, last_surf=N - 2)
    t0 = time.perf_counter()
    res = eng.trace_pupil_grid(fld, make_grid((-1., -1.), (1., 1.), num), 0, opts, want_pupil=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    R = num * num
    rows, r0 = 4, num // 2 - 2                          # rows through the middle of the pupil
    lo, hi = r0 * num, (r0 + rows) * num
    blk = eng.trace_pupil_grid(fld, make_grid((-1., -1.), (1., 1.), num, row_begin=r0, row_count=rows), 0, opts,
                               want_pupil=True)
    sa, sb = res.status[lo:hi].cpu().numpy(), blk.status[:rows * num].cpu().numpy()
    ok = sa == 0
    a = res.seg[:, :, lo:hi].cpu().numpy()[:, :, ok]
    b = blk.seg[:, :, :rows * num].cpu().numpy()[:, :, ok]
    same = bool(np.array_equal(sa, sb) and np.array_equal(a.view(np.int64), b.view(np.int64)) and
                np.array_equal(res.op[lo:hi].cpu().numpy()[ok].view(np.int64),
                               blk.op[:rows * num].cpu().numpy()[ok].view(np.int64)))
    print(json.dumps({'mode': 'FULL', 'num': num, 'rays': R, 'packet_GB': round(res._seg.numel() * 8 / 1e9, 1),
                      'largest_element_offset': int(res._seg.numel()), 'launch_ms_incl_allocation': ms,
                      'rows_compared': [r0, r0 + rows], 'rays_through_in_them': int(ok.sum()),
                      'rows_equal_their_row_block': same}))
    eng.close()


if __name__ == '__main__':
    if '--full' in sys.argv:
        full()
    else:
        main(int(sys.argv[1]) if len(sys.argv) > 1 else 65536)

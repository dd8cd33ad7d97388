#!/usr/bin/env python3
"""An interactive session in fast motion: thousands of model edits, each one a new surface table
(a new handle), a chief-ray aiming call and a small spot diagram through it.  Device memory,
pinned host memory and the process's resident set must stay flat.

    python tools/edit_loop_leak_check.py [edits]"""
import ctypes as C
import json
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(edits=3000):
    import torch
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi, workloads, SurfaceTable
    from rayoptics_amd.engine import TraceEngine, make_opts, make_grid
    wl = workloads.load('nikkor_c3')
    N = wl.n_ifcs
    opts = make_opts(flags=abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING,
                     out_mode=abi.OUT_HITS_COMPACT, first_surf=1, last_surf=N - 2, foc=wl.foc,
                     image_pt=wl.image_pts[0])
    grid = make_grid((-1., -1.), (1., 1.), 64)
    probs = []
    for m in wl.aim:
        a = abi.Aim()
        for i in range(3):
            a.pt0[i] = m['pt0'][i]
        a.z_enp, a.y_target, a.z_dir0 = m['z_enp'], 0.0, m['z_dir0']
        a.wvl_idx, a.surf, a.flip = m['wvl_idx'], m['surf'], 1
        probs.append(a)
    rows_t = type(wl.table.rows)
    samples = []
    t0 = time.time()
    for k in range(edits):
        rows = rows_t.from_buffer_copy(bytes(wl.table.rows))
        rows[3].cv = wl.table.rows[3].cv * (1.0 + 1e-6 * (k % 97))        # the edit
        tbl = SurfaceTable(rows, wl.table.n_table, wl.table.wvls, wl.table.stop_idx)
        eng = TraceEngine(tbl)
        eng.aim_chief_rays(probs)
        xy = eng.trace_pupil_grid_hits(wl.fields[0], grid, 0, opts)
        n = len(xy)
        del xy
        eng.close()
        if k % (edits // 10) == 0 or k == edits - 1:
            free, total = torch.cuda.mem_get_info()
            samples.append({'edit': k, 'device_used_MiB': (total - free) >> 20,
                            'rss_MiB': resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10, 'pairs': n})
    dev = [s['device_used_MiB'] for s in samples]
    rss = [s['rss_MiB'] for s in samples]
    print(json.dumps({'edits': edits, 'seconds': round(time.time() - t0, 1),
                      'ms_per_edit_cycle': (time.time() - t0) / edits * 1e3,
                      'device_used_MiB_first_last': [dev[1], dev[-1]], 'rss_MiB_first_last': [rss[1], rss[-1]],
                      'flat': bool(dev[-1] - dev[1] <= 8 and rss[-1] - rss[1] <= 64), 'samples': samples}))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3000)

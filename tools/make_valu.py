#!/usr/bin/env python3
"""profiles/valu_per_intersection.json from tools/pmc_collect.sh runs: wave-level VALU
instructions of the HITS (and FULL) trace kernel per ray-surface intersection, per workload.

    python tools/make_valu.py <tag> gpurun_out/pmc_<tag>_<workload> [...]

Each directory holds the counter CSVs of one workload's passes and the JSON line
tools/ab_bench.py printed (sq1.log: rays, intersections of one launch).  bench.py turns
the figure into a VALU issue fraction for the kernels it times live:
wave instructions x 4 cycles / (256 CUs x 4 SIMDs x 2.4 GHz x kernel time)."""
import json
import os
import subprocess
import sys


def ab_line(d):
    for name in ('sq1.log', 'sq2.log', 'write.log', 'fetch.log'):
        p = os.path.join(d, name)
        if not os.path.exists(p):
            continue
        for ln in open(p, errors='replace'):
            ln = ln.strip()
            if ln.startswith('{') and '"intersections"' in ln:
                return json.loads(ln)
    raise SystemExit(f'{d}: no ab_bench line with "intersections"')


def main(tag, dirs):
    out_path = os.path.join('profiles', 'valu_per_intersection.json')
    out = {}
    if os.path.exists(out_path):
        with open(out_path) as f:
            out = json.load(f)
    try:
        with open('ray-optics_amd/libroxtrace.so.srchash') as f:
            lib = f.read().strip()[:16]
    except OSError:
        lib = None
    for d in dirs:
        ab = ab_line(d)
        pm = json.loads(subprocess.check_output([sys.executable, 'tools/pmc_summary.py', d]))
        rec = {'valu_wave_insts_per_intersection': pm['HITS']['SQ_INSTS_VALU'] / ab['intersections'],
               'salu_wave_insts_per_intersection': pm['HITS']['SQ_INSTS_SALU'] / ab['intersections'],
               'full_valu_wave_insts_per_intersection': pm['FULL']['SQ_INSTS_VALU'] / ab['intersections'],
               'grid': f"{ab['num']}x{ab['num']} field {ab['field']}", 'intersections': ab['intersections'],
               'library_source_hash': lib,
               'source': f'{d} ({tag}): SQ_INSTS_VALU of trace_kernel<HITS> / intersections of one launch'}
        out[ab['workload']] = rec
        print(ab['workload'], rec)
        if 'HITS_FAST' in pm:       # the tolerance-mode twin (ROX_FAST_FP64), bench.py's roofline_hits_fast
            fr = {'valu_wave_insts_per_intersection': pm['HITS_FAST']['SQ_INSTS_VALU'] / ab['intersections'],
                  'salu_wave_insts_per_intersection': pm['HITS_FAST']['SQ_INSTS_SALU'] / ab['intersections'],
                  'grid': rec['grid'], 'intersections': ab['intersections'], 'library_source_hash': lib,
                  'source': f'{d} ({tag}): SQ_INSTS_VALU of trace_kernel<HITS, F_FAST> / intersections of one launch'}
            out[ab['workload'] + '_fast'] = fr
            print(ab['workload'] + '_fast', fr)
    with open(out_path, 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])

#!/usr/bin/env python3
"""profiles/traffic.json from a tools/pmc_collect.sh run: HBM bytes per FULL
launch = WRITE_SIZE + 2*FETCH_SIZE (KiB; FETCH_SIZE reads half the bytes of a
wide coalesced stream on gfx950 -- MI355X_MICROARCH.md section HBM)."""
import json
import subprocess
import sys


def lib_hash():
    """digest of the library build the counters were taken with (bench.py withholds the
    figure when the kernels have changed since)"""
    try:
        with open('ray-optics_amd/libroxtrace.so.srchash') as f:
            return f.read().strip()[:16]
    except OSError:
        return None


def main(pmc_dir, out, tag):
    j = json.loads(subprocess.check_output([sys.executable, 'tools/pmc_summary.py', pmc_dir]))
    full = j['FULL']
    hbm = full['WRITE_SIZE'] * 1024 + 2 * full['FETCH_SIZE'] * 1024
    d = {'workload': 'dblgauss_c2', 'num': 1024, 'kernel': 'trace_kernel<FULL,PUPIL>',
         'hbm_bytes_per_launch': hbm, 'WRITE_SIZE_KiB': full['WRITE_SIZE'],
         'FETCH_SIZE_KiB_uncorrected': full['FETCH_SIZE'], 'source': tag,
         'library_source_hash': lib_hash(),
         'note': 'separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), means over the '
                 'FULL-kernel dispatches of tools/ab_bench.py'}
    with open(out, 'w') as f:
        json.dump(d, f, indent=1)
    print(json.dumps(d))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else '')

#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel of libroxtrace.so, from
hipcc's -Rpass-analysis=kernel-resource-usage remarks (no GPU needed).

    python tools/kernel_resources.py [--only inst_radial] [extra hipcc flags ...] [--json out.json]
"""
import concurrent.futures
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'ray-optics_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math']
KEYS = {'VGPRs': 'vgpr', 'AGPRs': 'agpr', 'TotalSGPRs': 'sgpr',
        'ScratchSize [bytes/lane]': 'scratch', 'Occupancy [waves/SIMD]': 'occ',
        'LDS Size [bytes/block]': 'lds'}


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names),
                         capture_output=True, text=True).stdout.split('\n')
    return [o.replace('(anonymous namespace)::', '') for o in out]


def main():
    argv = sys.argv[1:]
    jout = None
    if '--json' in argv:
        i = argv.index('--json')
        jout = argv[i + 1]
        del argv[i:i + 2]
    only = None
    if '--only' in argv:
        i = argv.index('--only')
        only = argv[i + 1]
        del argv[i:i + 2]
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    if only:
        srcs = [s for s in srcs if only in os.path.basename(s)]

    def run(src):
        with tempfile.TemporaryDirectory() as td:
            p = subprocess.run(['hipcc', *FLAGS, *argv, '-c', src, '-o', os.path.join(td, 'r.o'),
                                '-Rpass-analysis=kernel-resource-usage'],
                               capture_output=True, text=True)
        if p.returncode:
            sys.stderr.write(p.stderr)
            raise SystemExit(p.returncode)
        return p.stderr

    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        text = '\n'.join(ex.map(run, srcs))
    rows, cur = [], None
    for line in text.split('\n'):
        m = re.search(r'remark: [^:]+:\d+:\d+: +(.*?) \[-Rpass-analysis', line)
        if not m:
            m = re.search(r':\d+:\d+: remark: +(.*?) \[-Rpass-analysis', line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith('Function Name:') or txt.startswith('Name:'):
            cur = {'name': txt.split(':', 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ':' in txt:
            k, v = txt.rsplit(':', 1)
            if k.strip() in KEYS:
                cur[KEYS[k.strip()]] = int(v)
    names = demangle([r['name'] for r in rows])
    for r, n in zip(rows, names):
        r['name'] = re.sub(r'\(.*\)$', '', n).replace('void ', '')
    rows.sort(key=lambda r: r['name'])
    print(f"{'kernel':60s} vgpr agpr sgpr scratch occ   lds")
    for r in rows:
        print(f"{r['name']:60s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('sgpr', 0):4d} "
              f"{r.get('scratch', 0):7d} {r.get('occ', 0):3d} {r.get('lds', 0):5d}")
    if jout:
        with open(jout, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Summarises the CSVs written by tools/pmc_collect.sh: per kernel, the mean of
every counter over its dispatches.  Prints JSON (kept under profiles/)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def main(d):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row['Kernel_Name']
                if 'trace_kernel' not in k:
                    continue
                short = 'FULL' if 'trace_kernel<0' in k else ('HITS' if 'trace_kernel<2' in k else k[:40])
                # trace_kernel<mode, gen, per-ray-wvl, FEAT, small>: FEAT & 64 = the tolerance-mode twin
                m = re.search(r'trace_kernel<\d+, \d+, \w+, (\d+)', k)
                if m and int(m.group(1)) & 64:
                    short += '_FAST'
                acc[short][row['Counter_Name']].append(float(row['Counter_Value']))
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
    for k, cs in acc.items():
        out[k]['dispatches'] = max(len(v) for v in cs.values())
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    main(sys.argv[1])

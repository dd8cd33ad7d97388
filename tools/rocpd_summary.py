#!/usr/bin/env python3
"""rocprofv3 (ROCm 7.2) writes a rocpd sqlite DB by default; this dumps its
`top_kernels` view (the --stats summary) as CSV for profiles/."""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    cur = c.execute('select name, total_calls, total_duration, average, percentage '
                    'from top_kernels')
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'pct'])
        for name, calls, tot, avg, pct in cur:
            if len(name) > 120:
                name = name[:117] + '...'
            w.writerow([name, calls, f'{tot:.3f}', f'{avg:.3f}', f'{pct:.2f}'])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])

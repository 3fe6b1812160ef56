#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats` sqlite result (rocpd) into the text summary kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, note=''):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    total = sum(r[2] for r in rows)
    with open(out_path, 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats summary (durations in us)\n')
        if note:
            f.write('# %s\n' % note)
        f.write('# total kernel time: %.1f us over %d dispatches\n' % (total, sum(r[1] for r in rows)))
        f.write('%-100s %8s %14s %12s %7s\n' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct'))
        for name, calls, tot, avg, pct in rows:
            short = name if len(name) <= 100 else name[:97] + '...'
            f.write('%-100s %8d %14.1f %12.2f %7.2f\n' % (short, calls, tot, avg, pct))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], ' '.join(sys.argv[3:]))

#!/usr/bin/env python
"""Ordered kernel list of the LAST repetition in a rocprofv3 --kernel-trace result (rocpd sqlite): start offset, duration
and gap to the previous kernel's end, for reading a captured step as a timeline.
usage: rocprof_sequence.py <db> <out.txt> <n_kernels_per_repetition | anchor kernel substring>"""
import sqlite3
import sys


def main(db_path, out_path, anchor):
    cur = sqlite3.connect(db_path).cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = 'kernels' if 'kernels' in tables else [t for t in tables if 'kernel' in t.lower()][0]
    cols = [r[1] for r in cur.execute('pragma table_info(%s)' % view)]
    rows = list(cur.execute('select name, start, end from %s order by start' % view))
    if anchor.isdigit():
        rows = rows[-int(anchor):]
    else:                                   # from the last launch whose name contains the anchor
        last = max(i for i, r in enumerate(rows) if anchor in r[0])
        prev = max(i for i, r in enumerate(rows[:last]) if anchor in r[0])
        rows = rows[prev + 1:last + 1]
    t0 = rows[0][1]
    with open(out_path, 'w') as f:
        f.write('# %d kernels, span %.1f us, busy %.1f us\n' % (len(rows), (rows[-1][2] - t0) / 1e3,
                                                               sum(r[2] - r[1] for r in rows) / 1e3))
        prev_end = t0
        for name, s, e in rows:
            f.write('%9.1f %8.1f %7.1f  %s\n' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:110]))
            prev_end = max(prev_end, e)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])

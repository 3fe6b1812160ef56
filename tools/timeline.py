"""GPU timeline of the pipelined step from a rocprofv3 kernel trace (rocpd sqlite): per HW queue busy time, the union of
all kernel intervals (GPU busy) and the idle share over the last steps.  usage: timeline.py <dir> [steps] [ms_per_step]"""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
qcol = 'queue_id' if 'queue_id' in cols else 'stream_id'
rows = list(cur.execute('select %s, start, end from %s order by start' % (qcol, kd)))
t_end = max(r[2] for r in rows)
window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
t0 = t_end - window_ms * 1e6
rows = [r for r in rows if r[1] >= t0]
span = (t_end - t0) / 1e6
byq = {}
for q, s, e in rows:
    byq.setdefault(q, []).append((s, e))
print('window %.1f ms, %d dispatches' % (span, len(rows)))
for q, iv in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e in iv) / 1e6
    gaps = sorted((iv[i + 1][0] - iv[i][1]) / 1e3 for i in range(len(iv) - 1))
    med = gaps[len(gaps) // 2] if gaps else 0
    print('queue %-6s %6d kernels  busy %7.2f ms (%5.1f %%)  avg kernel %6.1f us  median gap %6.1f us'
          % (q, len(iv), busy, 100 * busy / span, 1e3 * busy / len(iv), med))
ev = sorted([(s, 1) for _, s, e in rows] + [(e, -1) for _, s, e in rows])
depth, last, busy, hist = 0, t0, 0.0, {}
for t, d in ev:
    if depth > 0:
        busy += t - last
    hist[min(depth, 4)] = hist.get(min(depth, 4), 0) + (t - last)
    last = t
    depth += d
print('GPU busy (union) %.2f ms = %.1f %%' % (busy / 1e6, 100 * busy / 1e6 / span))
print('concurrency histogram (share of wall): ' + ', '.join('%d%s: %.1f%%' % (k, '+' if k == 4 else '', 100 * v / 1e6 / span)
                                                             for k, v in sorted(hist.items())))

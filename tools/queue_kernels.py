"""Per hardware queue: kernel counts per step from a rocprofv3 kernel trace (rocpd sqlite) of the pipelined bench.
usage: queue_kernels.py <dir> <steps-in-window> [window_ms]"""
import glob, sqlite3, sys, collections
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
namecol = 'display_name' if 'display_name' in cols else 'kernel_name'
names = dict(cur.execute('select id, %s from %s' % (namecol, ks)))
kcols = [r[1] for r in cur.execute('pragma table_info(%s)' % kd)]
qcol = 'queue_id' if 'queue_id' in kcols else 'stream_id'
rows = list(cur.execute('select %s, kernel_id, start, end from %s order by start' % (qcol, kd)))
steps = float(sys.argv[2])
t_end = max(r[3] for r in rows)
win = float(sys.argv[3]) if len(sys.argv) > 3 else None
if win:
    rows = [r for r in rows if r[2] >= t_end - win * 1e6]
byq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for q, k, s, e in rows:
    d = byq[q][names[k]]
    d[0] += 1
    d[1] += (e - s) / 1e3
for q, ks_ in byq.items():
    tot = sum(v[0] for v in ks_.values())
    print('== queue %s: %.0f kernels/step, %.2f ms/step of kernel time' % (q, tot / steps, sum(v[1] for v in ks_.values()) / steps / 1e3))
    for n, (c, us) in sorted(ks_.items(), key=lambda kv: -kv[1][0])[:28]:
        print('   %6.1f /step  avg %7.2f us  %s' % (c / steps, us / c, n[:100]))

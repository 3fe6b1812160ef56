"""Which kernels sit right before / after every __amd_rocclr_copyBuffer dispatch of the same queue (rocpd sqlite)?"""
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
print(kcols)
rows = list(cur.execute('select %s, kernel_id, start, end, grid_size_x, workgroup_size_x from %s order by start' % (qcol, kd)))
byq = collections.defaultdict(list)
for r in rows:
    byq[r[0]].append(r)
ctx = collections.Counter()
sizes = collections.Counter()
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if 'copyBuffer' in names[r[1]]:
            prev = names[rs[i - 1][1]][:60] if i else '-'
            nxt = names[rs[i + 1][1]][:60] if i + 1 < len(rs) else '-'
            ctx[(prev, nxt)] += 1
            sizes[(r[4], r[5])] += 1
for (p, n), c in ctx.most_common(25):
    print('%5d  after [%s]  before [%s]' % (c, p, n))
print('grid/block sizes:', sizes.most_common(8))

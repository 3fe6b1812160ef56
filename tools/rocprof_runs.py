"""Group consecutive identical dispatches of a rocprofv3 kernel trace (rocpd sqlite) and print their mean duration."""
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in cur.execute('pragma table_info(%s)' % ks)]
namecol = 'display_name' if 'display_name' in cols else 'kernel_name'
names = dict(cur.execute('select id, %s from %s' % (namecol, ks)))
rows = list(cur.execute('select kernel_id, grid_size_x, workgroup_size_x, start, end from %s order by start' % kd))
minrun = int(sys.argv[2]) if len(sys.argv) > 2 else 20
i = 0
while i < len(rows):
    j = i
    while j < len(rows) and rows[j][:3] == rows[i][:3]:
        j += 1
    if j - i >= minrun:
        d = [(r[4] - r[3]) / 1e3 for r in rows[i + 5:j]]
        print('%4d x %-70s grid %6d  avg %7.2f us  min %7.2f' % (j - i, names[rows[i][0]][:70], rows[i][1] // rows[i][2], sum(d) / len(d), min(d)))
    i = j

#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernel.sh <kernel-substring> <out.txt> <counter,counter,...> -- <command...>
# One rocprofv3 --pmc pass (<= 8 SQ counters) over a command; per-dispatch averages of each counter for one kernel.
kern=$1; out=$2; ctrs=$3; shift; shift; shift; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmck
( cd $root && timeout 600 rocprofv3 --kernel-trace --pmc ${ctrs//,/ } --output-format csv -d /tmp/pmck -o p -- "$@" > /tmp/pmck.log 2>&1 )
cd $root
python - "$kern" "$out" <<'PY'
import csv, glob, sys, collections
kern, out = sys.argv[1:3]
f = glob.glob('/tmp/pmck/**/*counter_collection.csv', recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    if kern in r['Kernel_Name']:
        k = r['Kernel_Name'][:60]
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
with open(out, 'w') as o:
    for k in tot:
        o.write(k + '\n')
        for c in sorted(tot[k]):
            o.write('  %-28s %16.0f  (avg of %d dispatches)\n' % (c, tot[k][c] / n[k][c], n[k][c]))
print(open(out).read())
PY

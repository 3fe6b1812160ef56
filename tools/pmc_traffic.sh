#!/bin/bash
# usage (GPU box, repo root): tools/pmc_traffic.sh <kernel-substring> <out.json>
# HBM traffic per launch of one kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short
# single-stream bench run, averaged over that kernel's dispatches; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md).
kern=$1; out=$2; short=$3; model=${4:-faces_objects}   # short: bench.py's name of the kernel (roofline.kernel)
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- \
    python $root/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc_$c.log 2>&1
done
cd $root
python - "$kern" "$out" "$short" "$model" <<'PY'
import csv, glob, json, sys
kern, out, short, model = sys.argv[1:5]
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('/tmp/pmc_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    tot, n = 0.0, 0
    name = None
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name'] and r['Counter_Name'] == c:
            tot += float(r['Counter_Value']); n += 1; name = r['Kernel_Name']
    res[c] = (tot / max(n, 1), n, name)
fetch_kb, n, name = res['FETCH_SIZE']
write_kb = res['WRITE_SIZE'][0]
j = {'command': 'rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --serial --steps 2 --warmup 1 '
                '--no-cpu-baseline --no-roofline (separate passes)',
     'kernel': short, 'model': model, 'kernel_symbol': name, 'dispatches': n, 'FETCH_SIZE_KB_per_launch_raw': round(fetch_kb, 1),
     'WRITE_SIZE_KB_per_launch': round(write_kb, 1),
     'gfx950_correction': 'FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads -> x2 '
                          '(MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected',
     'traffic_bytes_per_launch': int((2 * fetch_kb + write_kb) * 1024)}
json.dump(j, open(out, 'w'), indent=1)
print(json.dumps(j, indent=1))
PY

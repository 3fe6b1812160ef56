#!/bin/bash
# usage (GPU box, repo root): tools/pmc_generate_traffic.sh <beam> <out.json>
# HBM traffic of ONE captured decode step: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short
# generation run, summed over the kernels of the last complete step (between two bookkeeping launches);
# FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md).
beam=$1; out=$2
anchor=greedy_update; [ "$beam" != "1" ] && anchor=beam_update
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcg_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcg_$c -o p -- \
    python $root/bench.py --generate --beam $beam --steps 1 --warmup 2 > /tmp/pmcg_$c.log 2>&1
done
cd $root
python - "$anchor" "$out" "$beam" <<'PY'
import csv, glob, json, sys
anchor, out, beam = sys.argv[1:4]
res = {}
per = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('/tmp/pmcg_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == c]
    key = 'Dispatch_Id' if 'Dispatch_Id' in rows[0] else 'Correlation_Id'
    rows.sort(key=lambda r: int(r[key]))
    idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
    a, b = idx[-2], idx[-1]                      # the last complete step: (a, b]
    step = rows[a + 1:b + 1]
    res[c] = (sum(float(r['Counter_Value']) for r in step), len(step))
    per[c] = [(r['Kernel_Name'].split('(')[0][:60], float(r['Counter_Value'])) for r in step]
fetch_kb, n = res['FETCH_SIZE']
write_kb = res['WRITE_SIZE'][0]
j = {'command': 'rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --generate --beam %s --steps 1 '
                '--warmup 2 (separate passes), kernels of the last complete decode step' % beam,
     'beam': int(beam), 'kernels_per_step': n, 'FETCH_SIZE_KB_per_step_raw': round(fetch_kb, 1),
     'WRITE_SIZE_KB_per_step': round(write_kb, 1),
     'gfx950_correction': 'FETCH_SIZE counts 64 B per 128 B request for wide coalesced reads -> x2 '
                          '(MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected',
     'traffic_bytes_per_step': int((2 * fetch_kb + write_kb) * 1024),
     # kernel by kernel, in launch order: [name, MB fetched (x2 corrected), MB written]
     'per_kernel_MB': [[f[0], round(2 * f[1] / 1024, 2), round(w[1] / 1024, 2)] for f, w in zip(per['FETCH_SIZE'], per['WRITE_SIZE'])]}
json.dump(j, open(out, 'w'), indent=1)
print(json.dumps(j, indent=1))
PY

#!/bin/bash
# timeline of the kernels of the last timed steps of bench.py (start offsets and durations per queue): tools/step_timeline.sh OUT [bench args]
cd /tmp; export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu --no-host "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/tr/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last two recon launches bracket one whole step
rec = [i for i, r in enumerate(rows) if 'recon_chain' in r['Kernel_Name']]
m = len(rec) // 2 + 1
a, b = rec[m], rec[m + 2]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 1]:
    n = r['Kernel_Name'].split('(')[0].replace('lh264::', '')
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d < 0.05: continue
    print("q%-3s %8.2f ms  +%7.2f ms  %s" % (r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e6, d, n))
PY
rm -rf $OUT/tr
grep '^{' $OUT/bench.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"

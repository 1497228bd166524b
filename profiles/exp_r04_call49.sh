#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04ak
mkdir -p $O; rm -rf $O/prof_train_one
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_one -- python3 $R/bench.py --train --no-cpu --train-adam fused > $O/prof_train_one.log 2>&1 || echo "prof failed"
cd $R
tail -1 $O/prof_train_one.log | cut -c1-300
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/r04ak/prof_train_one/*/*_kernel_stats.csv'))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('total kernel ms', round(tot / 1e6, 1), 'launches', calls)
for r in rows[:14]:
    print('  ', r['Name'][:72].ljust(72), r['Calls'], round(float(r['AverageNs']) / 1e3, 1), round(100 * float(r['TotalDurationNs']) / tot, 1))
PY

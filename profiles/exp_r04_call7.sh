#!/bin/bash
# Round 4, seventh GPU call: 2 vs 3 pipeline streams for the small legs on the one-launch form.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04g
mkdir -p $O
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms\" for k,v in d['configs'].items()), round(d.get('value_incl_d2h',0)/1e6,1))"; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), 'clock', [round(c,3) for c in d['clock_ghz']])"; }
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --warmup 5 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained --no-serial-check --warmup 5"
$B --steps 10 > /dev/null 2>&1
for i in 1 2; do
echo "legs 2 streams: $(STTODE_LAGGED=2 $L 2>/dev/null | legs)" | tee -a $O/streams.txt
echo "legs 3 streams: $(STTODE_LAGGED=3 $L 2>/dev/null | legs)" | tee -a $O/streams.txt
done
for sc in 64 128 256 512; do
echo "scenes $sc 2 streams: $(STTODE_LAGGED=2 $B --steps 40 --scenes $sc 2>/dev/null | line)" | tee -a $O/streams.txt
echo "scenes $sc 3 streams: $(STTODE_LAGGED=3 $B --steps 40 --scenes $sc 2>/dev/null | line)" | tee -a $O/streams.txt
done

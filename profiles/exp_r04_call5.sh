#!/bin/bash
# Round 4, fifth GPU call: the call as ONE launch (front-end in the roles, latents in the roles, metrics in the groups) + workers on all slots.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04e
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(r.get('frac',0),3), 'incl d2h', round(d['value_incl_d2h']/1e6,2), d.get('ade_fde_synthetic'))"; }
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms\" for k,v in d['configs'].items()), round(d.get('value_incl_d2h',0)/1e6,1))"; }
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained --no-serial-check --warmup 5"
$B --steps 10 > /dev/null 2>$O/first.err || { tail -20 $O/first.err; exit 1; }
for i in 1 2; do
for st in 20 80 160; do
echo "steps $st all-in-one, workers 512     : $($B --steps $st 2>/dev/null | line)" | tee -a $O/ab.txt
echo "steps $st all-in-one, workers 444     : $(STTODE_LAG_WORKERS=444 $B --steps $st 2>/dev/null | line)" | tee -a $O/ab.txt
echo "steps $st all-in-one, one per group   : $(STTODE_LAG_WORKERS=0 $B --steps $st 2>/dev/null | line)" | tee -a $O/ab.txt
echo "steps $st bok kernel (no fused metrics): $(STTODE_FUSED_METRICS=0 $B --steps $st 2>/dev/null | line)" | tee -a $O/ab.txt
echo "steps $st front-end launches           : $(STTODE_LAG_FE=0 $B --steps $st 2>/dev/null | line)" | tee -a $O/ab.txt
done; done
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --warmup 5 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
for i in 1 2; do
echo "legs all-in-one, workers 512   : $($L 2>/dev/null | legs)" | tee -a $O/ab.txt
echo "legs all-in-one, one per group : $(STTODE_LAG_WORKERS=0 $L 2>/dev/null | legs)" | tee -a $O/ab.txt
done
for sc in 128 256 1024; do
echo "scenes $sc workers 512   : $($B --steps 40 --scenes $sc 2>/dev/null | line)" | tee -a $O/ab.txt
echo "scenes $sc one per group : $(STTODE_LAG_WORKERS=0 $B --steps 40 --scenes $sc 2>/dev/null | line)" | tee -a $O/ab.txt
done
STTODE_HIP_LIB=$R/sttode_amd/lib/variants/lib_trace.so TRACE_NAME=allinone timeout -k 10 300 python profiles/exp_r03_trace.py 512 24 > $O/trace_allinone_512.txt 2>&1 || { tail -20 $O/trace_allinone_512.txt; exit 1; }
tail -7 $O/trace_allinone_512.txt

#!/bin/bash
# Round 4, eighth GPU call: full GPU suite on the current code + legs (NBA front-end inside the embedding launch) + smoke.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04h
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms/path{v['roofline']['path_frac_executed']:.3f}\" for k,v in d['configs'].items()), 'd2h', round(d.get('value_incl_d2h',0)/1e6,1), 'frac', round(d['roofline']['frac'],3), 'path', round(d['roofline']['path_frac_executed'],3))"; }
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --warmup 5 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
for i in 1 2; do
echo "legs, NBA front-end in the embedding launch: $($L 2>/dev/null | legs)" | tee -a $O/legs.txt
echo "legs, front-end launches (STTODE_LAG_FE=0)  : $(STTODE_LAG_FE=0 $L 2>/dev/null | legs)" | tee -a $O/legs.txt
done

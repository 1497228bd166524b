#!/bin/bash
# Round 4, call 29: role waves at s_setprio 3 (they outlast the groups in small launches) -- legs and headline, alternating.
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04ac
mkdir -p $O
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms\" for k,v in d['configs'].items()))"; }
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --warmup 5 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
for i in 1 2 3; do
echo "prio 0: $($L 2>/dev/null | legs)" | tee -a $O/role_prio_ab.txt
echo "prio 3: $(STTODE_ROLE_PRIO=3 $L 2>/dev/null | legs)" | tee -a $O/role_prio_ab.txt
done

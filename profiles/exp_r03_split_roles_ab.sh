#!/bin/bash
# Same-box alternating A/B of split per-agent roles (five workgroups per 16-agent tile: E | G | three tables; STTODE_ROLE_LEAD=-2,
# sttode_set_fused mode 4) against one role workgroup per tile (STTODE_ROLE_LEAD=-1, the default): headline, serial launch, the four legs.
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(r.get('frac',0),3), round(r.get('frac_serial_equivalent',0),3))"; }
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms\" for k,v in d['configs'].items()))"; }
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained"
$B --steps 10 > /dev/null 2>&1
for i in 1 2 3; do
for st in 20 40; do
echo "steps $st split : $(STTODE_ROLE_LEAD=-2 $B --steps $st 2>/dev/null | line)"
echo "steps $st single: $(STTODE_ROLE_LEAD=-1 $B --steps $st 2>/dev/null | line)"
done; done
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
for i in 1 2 3; do
echo "legs split : $(STTODE_ROLE_LEAD=-2 $L 2>/dev/null | legs)"
echo "legs single: $(STTODE_ROLE_LEAD=-1 $L 2>/dev/null | legs)"
done

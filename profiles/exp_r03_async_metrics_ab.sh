#!/bin/bash
# Same-box alternating A/B: best-of-K metrics on the call's own pipeline stream (model.best_of_k_async, bench.py --async-metrics) against
# the kernel on the caller's stream (default): headline at 20 / 40 / 80 steps, the four legs.
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(r.get('frac',0),3), d.get('ade_fde_synthetic'))"; }
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms\" for k,v in d['configs'].items()), round(d.get('value_incl_d2h',0)/1e6,1))"; }
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained --no-serial-check"
$B --steps 10 > /dev/null 2>&1
for i in 1 2 3; do
for st in 20 40 80; do
echo "steps $st async: $($B --steps $st --async-metrics 2>/dev/null | line)"
echo "steps $st sync : $($B --steps $st 2>/dev/null | line)"
done; done
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
for i in 1 2 3; do
echo "legs async: $($L --async-metrics 2>/dev/null | legs)"
echo "legs sync : $($L 2>/dev/null | legs)"
done

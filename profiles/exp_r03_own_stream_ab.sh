#!/bin/bash
# Same-box alternating A/B: H2D, latents, launch and metrics of a pipelined call all on the call's own pipeline stream (bench.py --own-stream
# --depth 3; model.next_async_stream) against the default step (inputs and metrics on the caller's stream, depth 4).
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(r.get('frac',0),3), d.get('ade_fde_synthetic'))"; }
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms\" for k,v in d['configs'].items()), round(d.get('value_incl_d2h',0)/1e6,1))"; }
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained --no-serial-check --warmup 5"
$B --steps 10 > /dev/null 2>&1
for i in 1 2 3; do
for st in 20 40 80; do
echo "steps $st own-stream: $($B --steps $st --own-stream --depth 3 2>/dev/null | line)"
echo "steps $st default   : $($B --steps $st 2>/dev/null | line)"
done; done
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --warmup 5 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
for i in 1 2; do
echo "legs own-stream: $($L --own-stream --depth 3 2>/dev/null | legs)"
echo "legs default   : $($L 2>/dev/null | legs)"
done

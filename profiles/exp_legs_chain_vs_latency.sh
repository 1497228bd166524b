#!/bin/bash
# Secondary legs of the bench with the per-trajectory stage forced to the fused chain (-1 = automatic threshold) or to the three-kernel
# form (0), plus the host's enqueue time per step:  gpurun -- 'bash profiles/exp_legs_chain_vs_latency.sh'
for C in -1 0; do
  echo "STTODE_CHAIN=$C"
  STTODE_CHAIN=$C timeout -k 10 300 python bench.py --no-cpu --no-train --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', round(d['value']/1e6,2), round(d['ms_per_step'],3), 'host', round(d.get('host_enqueue_ms_per_step',0),3))
for k,v in d.get('configs',{}).items(): print(k, round(v['value']/1e6,2), round(v['ms_per_step'],3), 'host', round(v.get('host_enqueue_ms_per_step',0),3), v.get('roofline',{}).get('kernel'), v.get('roofline',{}).get('frac'))
"
done

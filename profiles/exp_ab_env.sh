#!/bin/bash
# Same-box A/B of one environment switch on the headline bench, alternating runs (the first process on a fresh box runs slow: discarded):
#   gpurun -- 'bash profiles/exp_ab_env.sh STTODE_GRU0_LAT_TILES 1 4096 "512 256"'
VAR=$1; A=$2; B=$3; SCENES=${4:-512}
timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 > /dev/null 2>&1
for S in $SCENES; do
  for i in 1 2 3 4; do
    for V in $A $B; do
      echo "$VAR=$V scenes=$S: $(env $VAR=$V timeout -k 10 200 python bench.py --legs none --no-cpu --scenes $S --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(d['roofline']['frac'],3))")"
    done
  done
done

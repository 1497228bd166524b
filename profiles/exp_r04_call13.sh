#!/bin/bash
# Round 4, thirteenth GPU call: fragment look-ahead in the roles' tile feed -- lagged tests, headline at 20 / 80 steps, legs, small batches.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04m
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "lagged or headline or latents or fused_metrics or zero_copy or mixed" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-serial-check --warmup 5"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), 'sustained', round(d['sustained']['value']/1e6,2) if 'sustained' in d else None)"; }
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms/path{v['roofline']['path_frac_executed']:.3f}\" for k,v in d['configs'].items()))"; }
for i in 1 2 3; do echo "512 scenes, 20 steps + sustained: $($B --steps 20 2>/dev/null | line)" | tee -a $O/prefetch.txt; done
for sc in 64 128 256; do echo "scenes $sc: $($B --steps 40 --scenes $sc --no-sustained 2>/dev/null | line)" | tee -a $O/prefetch.txt; done
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --warmup 5 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
for i in 1 2; do echo "legs: $($L 2>/dev/null | legs)" | tee -a $O/prefetch.txt; done

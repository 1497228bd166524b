#!/bin/bash
# Round 4, twelfth GPU call: the attention kernel with its columns split over the four waves -- full GPU suite, the integrator timing, NBA legs.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04l
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 300 python profiles/exp_r04_ode.py 512 2>&1 | grep "per call" | tee $O/ode_config5.txt
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms/path{v['roofline']['path_frac_executed']:.3f}\" for k,v in d['configs'].items()))"; }
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --warmup 5 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
for i in 1 2; do echo "legs: $($L 2>/dev/null | legs)" | tee -a $O/legs.txt; done

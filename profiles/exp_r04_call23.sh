#!/bin/bash
# Round 4, call 23: GRU gradient products grouped; 4 pipeline streams against 3 on the legs and the headline.
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04w
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q -k "training_step or lagged" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
for i in 1 2; do echo "nba-size step (GRU products grouped): $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_step.txt; done
legs() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']/1e6,2), ' '.join(f\"{k}={v['value']/1e6:.1f}M/{v['ms_per_step']:.3f}ms/path{v['roofline']['path_frac_executed']:.3f}\" for k,v in d['configs'].items()))"; }
L="timeout -k 10 300 python bench.py --no-cpu --no-train --steps 20 --warmup 5 --no-exploratory --no-per-scene --no-sustained --no-serial-check"
for i in 1 2; do
echo "3 streams: $($L 2>/dev/null | legs)" | tee -a $O/streams_3_vs_4.txt
echo "4 streams: $(STTODE_LAGGED=4 $L 2>/dev/null | legs)" | tee -a $O/streams_3_vs_4.txt
done

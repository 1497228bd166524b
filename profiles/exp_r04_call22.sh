#!/bin/bash
# Round 4, call 22: grouped launches (decoder_x / decoder_y layer by layer), headline with the clock pre-warm.
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04v
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q -k "training or tlinear or sampler or train_ or layer_backward or grouped" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
for i in 1 2; do
echo "paired decoder MLPs, grouped launches: $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_group_ab.txt
echo "one product (pair) per launch (STTODE_TRAIN_PAIRED=0): $(STTODE_TRAIN_PAIRED=0 timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_group_ab.txt
done
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-serial-check --warmup 5 --steps 20"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), d['clock_ghz'], 'sustained', round(d['sustained']['value']/1e6,2) if 'sustained' in d else None)"; }
for i in 1 2; do
echo "clock pre-warm 40: $($B 2>/dev/null | line)" | tee -a $O/prewarm_ab.txt
echo "clock pre-warm  0: $($B --clock-prewarm 0 2>/dev/null | line)" | tee -a $O/prewarm_ab.txt
done

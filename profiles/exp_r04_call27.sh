#!/bin/bash
# Round 4, call 27: grouped launches at scene sizes (decoder_x / decoder_y layers, the two encoder trunks' backward walked together).
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04aa
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "training or tlinear or sampler or train_ or layer_backward or grouped or stale" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
T="timeout -k 10 300 python bench.py --train --no-cpu"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), 'foreach', round(d.get('ms_per_step_foreach_adam',0),4))"; }
for i in 1 2; do
echo "grouped at scene sizes        : $($T 2>/dev/null | line)" | tee -a $O/train_scene_group_ab.txt
echo "STTODE_TRAIN_PAIRED=0         : $(STTODE_TRAIN_PAIRED=0 $T 2>/dev/null | line)" | tee -a $O/train_scene_group_ab.txt
done
for i in 1 2; do echo "nba-size step: $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_scene_group_ab.txt; done

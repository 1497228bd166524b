#!/bin/bash
# Round 4, call 14: NBA-size training step replayed as a hipGraph (threshold 100 -> 512 agents), every native call of that step timed alone,
# host profile of the one-scene-per-call loop.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04n
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "training_step or tlinear or batched_scene_training" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
for i in 1 2; do
echo "graph replay : $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_graph_ab.txt
echo "eager (<=100): $(STTODE_TRAIN_GRAPH_MAX=100 timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_graph_ab.txt
done
timeout -k 10 300 python profiles/exp_r04_train_shapes.py > $O/train_shapes.txt 2>&1 || tail -20 $O/train_shapes.txt
head -50 $O/train_shapes.txt
timeout -k 10 200 python profiles/exp_r03_host_profile.py > $O/per_scene_host_profile.txt 2>&1
head -30 $O/per_scene_host_profile.txt | cut -c1-150

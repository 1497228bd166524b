#!/bin/bash
# Round 4, call 15: a layer's backward at batch sizes as ONE launch (dX tiles + dW tiles x splits), split-sum reductions deferred to one
# launch per 16 gradients.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04o
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q -k "training or tlinear or sampler or train_" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
for i in 1 2; do
echo "fused backward launch + deferred reductions: $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_bwd_ab.txt
echo "two launches per layer (STTODE_TGEMM_BWD=0): $(STTODE_TGEMM_BWD=0 timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_bwd_ab.txt
done
timeout -k 10 300 python profiles/exp_r04_train_shapes.py > $O/train_shapes.txt 2>&1 || tail -20 $O/train_shapes.txt
head -24 $O/train_shapes.txt

#!/bin/bash
# Round 4, call 16: epilogues of the training GEMMs on 16-byte pieces (all operands requested before the first is used).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04p
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q -k "training or tlinear or sampler or train_" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
for i in 1 2; do
echo "nba-size step: $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_step.txt
done
timeout -k 10 300 python profiles/exp_r04_train_shapes.py > $O/train_shapes.txt 2>&1 || tail -20 $O/train_shapes.txt
head -24 $O/train_shapes.txt
timeout -k 10 300 python bench.py --legs none --no-cpu --no-exploratory --no-per-scene --no-serial-check --no-sustained --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train', json.dumps({k: d['train'][k] for k in ('ms_per_step','ms_per_step_foreach_adam') if k in d['train']}))" | tee -a $O/train_step.txt

#!/bin/bash
# Round 4, call 19: full GPU suite on the training changes + the default bench line.
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04s
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],3), 'frac', d['roofline']['frac'])
print('train', {k: d['train'][k] for k in d['train'] if k.startswith('ms_per') or k=='value'})
print('per_scene', d['per_scene']['ms_per_scene'])
"
for i in 1 2; do echo "nba-size step: $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_step.txt; done

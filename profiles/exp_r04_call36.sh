#!/bin/bash
# Round 4, call 36: the LDS-tiled GEMM from 1024 columns on (the GRU projections of a one-scene step: 1680 step-columns) instead of 2048.
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04ae
mkdir -p $O
STTODE_TGEMM_MIN_COLS=1024 timeout -k 10 900 python -m pytest tests -m gpu -q -k "training_step or tlinear or layer_backward" > $O/gputests_1024.log 2>&1; tail -4 $O/gputests_1024.log
T="timeout -k 10 300 python bench.py --train --no-cpu"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), 'foreach', round(d.get('ms_per_step_foreach_adam',0),4))"; }
for i in 1 2; do
echo "tgemm above 2048 columns: $($T 2>/dev/null | line)" | tee -a $O/tgemm_min_cols_ab.txt
echo "tgemm above 1024 columns: $(STTODE_TGEMM_MIN_COLS=1024 $T 2>/dev/null | line)" | tee -a $O/tgemm_min_cols_ab.txt
done

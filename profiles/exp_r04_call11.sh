#!/bin/bash
# Round 4, eleventh GPU call: zero-copy futures (the lagged launch writes the predictions straight to pinned host memory).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04k
mkdir -p $O
timeout -k 10 100 python profiles/tmp/pin_test.py 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "zero_copy or lagged or headline or fused_metrics" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained --no-serial-check --warmup 5"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'incl d2h', round(d['value_incl_d2h']/1e6,2), round(d['ms_per_step_incl_d2h'],3))"; }
for i in 1 2; do
echo "futures written to pinned host memory by the launch: $($B --steps 40 2>/dev/null | line)" | tee -a $O/d2h_ab.txt
echo "d2h copy on the call's stream                       : $(STTODE_BENCH_D2H=own $B --steps 40 2>/dev/null | line)" | tee -a $O/d2h_ab.txt
done

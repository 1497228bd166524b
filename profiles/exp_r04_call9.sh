#!/bin/bash
# Round 4, ninth GPU call: new robustness test, the lagged stress run, D2H placement A/B.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04i
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "mixed_batch or lagged or latents or fused_metrics" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 400 python profiles/exp_r04_stress.py 3000 512 f32 2>&1 | tail -3 | tee $O/stress_lagged.txt
timeout -k 10 300 python profiles/exp_r04_stress.py 4000 96 f32 2>&1 | tail -2 | tee -a $O/stress_lagged.txt
timeout -k 10 400 python profiles/exp_r04_stress.py 1500 512 bf16x3 2>&1 | tail -2 | tee -a $O/stress_lagged.txt
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained --no-serial-check --warmup 5"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'incl d2h', round(d['value_incl_d2h']/1e6,2), round(d['ms_per_step_incl_d2h'],3))"; }
for i in 1 2; do
echo "d2h on the call's stream : $($B --steps 40 2>/dev/null | line)" | tee -a $O/d2h_ab.txt
echo "d2h on a copy stream     : $(STTODE_BENCH_D2H=copy $B --steps 40 2>/dev/null | line)" | tee -a $O/d2h_ab.txt
done

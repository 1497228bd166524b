#!/bin/bash
# Round 4, first GPU call: baseline of the round-3 code on this round's box + the evidence the review asked for:
#   (1) GPU tests, (2) same-box 20 / 80-step A/B of the default and own-stream step forms, (3) rocprofv3 --kernel-trace of an 80-step run of
#   each form (launch cadence: profiles/summarize_cadence.py), (4) the groups-only upper bound (profiles/exp_r04_groups_only.py).
#   gpurun --timeout 1100 -- bash profiles/exp_r04_call1.sh
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04a
mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -5 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(r.get('frac',0),3))"; }
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained --no-serial-check --warmup 5"
$B --steps 10 > /dev/null 2>&1
for i in 1 2; do
for st in 20 80 160; do
echo "steps $st own-stream: $($B --steps $st --own-stream --depth 3 2>/dev/null | line)" | tee -a $O/ab_forms.txt
echo "steps $st default   : $($B --steps $st 2>/dev/null | line)" | tee -a $O/ab_forms.txt
done; done
cd /tmp
BP="$R/bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained --no-serial-check --warmup 5 --time-every 0"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr_default -o tr -- python3 $BP --steps 80 > $O/tr_default.log 2>&1 \
  && python3 $R/profiles/summarize_cadence.py $O/tr_default/tr_results.db > $O/cadence_default_80.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr_own -o tr -- python3 $BP --steps 80 --own-stream --depth 3 > $O/tr_own.log 2>&1 \
  && python3 $R/profiles/summarize_cadence.py $O/tr_own/tr_results.db > $O/cadence_own_stream_80.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tr_default20 -o tr -- python3 $BP --steps 20 > $O/tr_default20.log 2>&1 \
  && python3 $R/profiles/summarize_cadence.py $O/tr_default20/tr_results.db > $O/cadence_default_20.txt || exit 1
cd $R
for a in "512 40 3 0" "512 40 2 0" "512 40 3 26" "512 40 2 26" "256 40 3 0" "128 60 3 0"; do
  timeout -k 10 200 python profiles/exp_r04_groups_only.py $a 2>&1 | grep groups-only | tee -a $O/groups_only.txt
done

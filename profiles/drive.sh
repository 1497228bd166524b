#!/bin/bash
# Named drivers for the one-GPU box (round 5 housekeeping: rounds 3-4 kept one throw-away script per gpurun call -- 49 `exp_r04_call*.sh` --
# whose bodies were always one of the five sequences below with another environment switch; the result files under profiles/r0N/ name the
# switch they compared, `git log -- profiles/exp_r04_call*.sh` keeps the originals).
#
#   gpurun -- 'bash profiles/drive.sh tests  OUT ["-k expr"]'           GPU test suite (or a subset) -> gpurun_out/OUT/gputests.log
#   gpurun -- 'bash profiles/drive.sh ab     OUT VAR A B ["bench flags"]'   alternating same-box A/B of ONE environment switch on bench.py (4 pairs)
#   gpurun -- 'bash profiles/drive.sh train  OUT [VAR A B]'              NBA-size step (32 x 11) and the one-scene train line, optionally as an A/B
#   gpurun -- 'bash profiles/drive.sh bench  OUT ["bench flags"]'        one bench line -> gpurun_out/OUT/bench.json (+ a short summary on stdout)
#   gpurun -- 'bash profiles/drive.sh prof   OUT NAME -- prog args..'    rocprofv3 --kernel-trace --stats of a program -> gpurun_out/OUT/NAME_kernel_stats.csv
set -o pipefail
export TMPDIR=/tmp
MODE=$1; OUT=$PWD/gpurun_out/$2; shift 2
mkdir -p "$OUT"
short() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'M traj/s', round(d['ms_per_step'],3), 'ms', 'frac', round(d['roofline']['frac'],3) if d.get('roofline') else '')"; }
trainline() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one-scene step', round(d['ms_per_step'],4), 'ms', d.get('optimizer'), '|', round(d.get('ms_per_step_fused_adam',0),4), 'ms torch fused |', round(d.get('ms_per_step_foreach_adam',0),4), 'ms torch foreach')"; }
case $MODE in
tests)
    timeout -k 10 1100 python -m pytest tests -m gpu -q $1 > "$OUT/gputests.log" 2>&1; tail -5 "$OUT/gputests.log" ;;
ab)
    VAR=$1; A=$2; B=$3; FLAGS=${4:---legs none --no-cpu --no-train --no-per-scene --no-exploratory --no-sustained --steps 40}
    timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-per-scene --no-exploratory --no-sustained --steps 10 > /dev/null 2>&1   # (the first process on a fresh box runs slow)
    for i in 1 2 3 4; do for V in "$A" "$B"; do
        echo "$VAR=$V: $(env $VAR=$V timeout -k 10 300 python bench.py $FLAGS 2>/dev/null | short)" | tee -a "$OUT/ab_$VAR.txt"
    done; done ;;
train)
    run() { echo "$1 nba-size step: $(env $2 timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a "$OUT/train_step.txt"
            echo "$1 $(env $2 timeout -k 10 300 python bench.py --train --no-cpu 2>/dev/null | trainline)" | tee -a "$OUT/train_step.txt"; }
    if [ -n "$1" ]; then for i in 1 2; do run "$1=$2" "$1=$2"; run "$1=$3" "$1=$3"; done; else run "" "X_=1"; run "" "X_=1"; fi ;;
bench)
    timeout -k 10 1100 python bench.py $1 > "$OUT/bench.json" 2> "$OUT/bench.err"; short < "$OUT/bench.json" ;;
prof)
    NAME=$1; shift 2
    R=$PWD; ARGS=(); for a in "$@"; do if [ -e "$R/$a" ]; then ARGS+=("$R/$a"); else ARGS+=("$a"); fi; done; set -- "${ARGS[@]}"   # (rocprofv3 runs from /tmp: repo files by absolute path)
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$NAME" -o "$NAME" -- "$@" > "$OUT/prof_$NAME.log" 2>&1
    cd "$R"
    f=$(find "$OUT/prof_$NAME" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${NAME}_kernel_stats.csv" && head -12 "$OUT/${NAME}_kernel_stats.csv" | cut -c1-160 ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac

#!/bin/bash
# Round-4 artefacts of the round's LAST code (second half: training kernels, host path): full GPU suite, smoke, the default bench line,
# rocprofv3 --kernel-trace --stats of the pipelined / serial headline and of the training steps.  gpurun --timeout 1190 -- 'bash profiles/collect_r04b.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04colb
mkdir -p $O
rm -rf $O/prof_*
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err || { echo "bench failed"; tail -5 $O/final_bench.err; exit 1; }
BQ="--legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial_headline -- python3 $R/bench.py --serial $BQ --steps 20 --warmup 3 > $O/prof_serial_headline.log 2>&1 || echo "serial prof failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipelined_headline -- python3 $R/bench.py $BQ --steps 40 --warmup 5 --no-serial-check > $O/prof_pipelined_headline.log 2>&1 || echo "pipelined prof failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_nba -- python3 $R/profiles/exp_train_nba_profile.py > $O/prof_train_nba.log 2>&1 || echo "train prof failed"
cd $R
for i in 1 2; do echo "nba-size step: $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_step.txt; done
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04colb/final_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('frac_serial_equivalent'), 'cpu', d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline'])
print('incl d2h', d['value_incl_d2h'], 'sustained', d['sustained']['value'], 'per_scene', d['per_scene']['ms_per_scene'], 'train', d['train']['ms_per_step'], d['train'].get('ms_per_step_foreach_adam'))
print({k: round(v['value'] / 1e6, 1) for k, v in d['configs'].items()})
PY

#!/bin/bash
# Round-5 measurement artefacts of the round's last code (run through gpurun from the repo root, raw output under gpurun_out/r05col; the
# summaries are copied into profiles/r05 afterwards):   gpurun --timeout 1190 -- 'bash profiles/collect_r05.sh [part]'
#   part a: CPU suite on the BOX'S HOST (review item 7), GPU suite, smoke, the driver's bench line, kernel stats of the serial / pipelined headline
#   part b: counters on the lagged launch (separate --pmc passes), training kernel stats + step timings, NBA evaluation rate, clock pre-warm A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r05col
mkdir -p $O
BQ="--legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained"
part=${1:-a}
if [ "$part" = a ]; then
  rm -rf $O/prof_*
  timeout -k 10 600 python -m pytest tests -q -m "not gpu" > $O/cputests_on_gpu_box_host.log 2>&1; tail -2 $O/cputests_on_gpu_box_host.log
  timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
  tail -2 $O/gputests.log
  timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
  tail -1 $O/smoke.log
  timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err || { echo "bench failed"; tail -5 $O/final_bench.err; exit 1; }
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial_headline -- python3 $R/bench.py --serial $BQ --steps 20 --warmup 3 > $O/prof_serial_headline.log 2>&1 || echo "serial prof failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipelined_headline -- python3 $R/bench.py $BQ --steps 40 --warmup 5 --no-serial-check > $O/prof_pipelined_headline.log 2>&1 || echo "pipelined prof failed"
  cd $R
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05col/final_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('frac_serial_equivalent'), 'cpu', d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline'])
print('parity', d['parity'])
print('incl d2h', d['value_incl_d2h'], 'sustained', d['sustained']['value'], 'per_scene', d['per_scene']['ms_per_scene'], 'train', d['train']['ms_per_step'], d['train'].get('ms_per_step_foreach_adam'))
print({k: (round(v['value'] / 1e6, 1), round(v['roofline']['path_frac_executed'], 3), v['parity']['max_err_over_1_plus_abs_ref']) for k, v in d['configs'].items()})
PY
else
  rm -rf $O/pmc_* $O/prof_train*
  cd /tmp
  for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    T=$(echo $P | cut -d" " -f1)
    timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$T -- python3 $R/bench.py $BQ --steps 8 --warmup 4 --no-serial-check > $O/pmc_$T.log 2>&1 || { echo "pmc $T failed"; }
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_nba -- python3 $R/profiles/exp_train_nba_profile.py > $O/prof_train_nba.log 2>&1 || echo "train prof failed"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_one_scene -- python3 $R/bench.py --train --no-cpu > $O/prof_train_one_scene.log 2>&1 || echo "one-scene train prof failed"
  cd $R
  for i in 1 2; do echo "nba-size step: $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_step.txt; done
  timeout -k 10 300 python profiles/exp_r05_nba_eval_rate.py 2>/dev/null | tee $O/nba_eval_rate.txt
  # clock pre-warm of the contract region: 6 alternating pairs (review item 8: round 4 had 2, contradictory)
  for i in 1 2 3 4 5 6; do for V in 0 40; do
    echo "prewarm $V: $(timeout -k 10 200 python bench.py $BQ --no-serial-check --steps 20 --warmup 5 --prewarm $V 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],4), [round(c,3) for c in d['clock_ghz']])")" | tee -a $O/ab_clock_prewarm.txt
  done; done
fi

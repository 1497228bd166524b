#!/bin/bash
# Round-4 measurement artefacts (run through gpurun from the repo root; raw output under gpurun_out/r04c, summaries are copied into
# profiles/r04 by `python profiles/summarize_pmc.py gpurun_out/r04col profiles/r04` in the authoring container):
#   gpurun --timeout 1150 -- 'bash profiles/collect_r04.sh'
# PMC counters are collected in their own passes (never combined with sys/hip traces), the program directly after "--".
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04col
mkdir -p $O
rm -rf $O/prof_* $O/pmc_* $O/tl_*
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/final_bench.json 2> $O/final_bench.err || { echo "bench failed"; tail -5 $O/final_bench.err; exit 1; }
BQ="--legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-sustained"
cd /tmp
# (1) SERIAL kernel stats, no counters: flop_per_launch / AverageNs / 157.3e12 is the plain per-launch roofline fraction of a serial call
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial_headline -- python3 $R/bench.py --serial $BQ --steps 20 --warmup 3 > $O/prof_serial_headline.log 2>&1 || { echo "serial prof failed"; exit 1; }
# (2) PIPELINED (product path) kernel stats + the union of the launch intervals from the trace's own timestamps
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipelined_headline -- python3 $R/bench.py $BQ --steps 40 --warmup 5 --no-serial-check > $O/prof_pipelined_headline.log 2>&1 || echo "pipelined prof failed"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tl_pipelined -o tl -- python3 $R/bench.py $BQ --steps 40 --warmup 5 --no-serial-check > $O/tl_pipelined.log 2>&1 \
  && python3 $R/profiles/summarize_timeline.py $O/tl_pipelined/tl_results.db traj_chain 24 8 > $O/timeline_pipelined.txt \
  && python3 $R/profiles/summarize_cadence.py $O/tl_pipelined/tl_results.db > $O/cadence_pipelined.txt
for L in ucy_2048 sdd_1024 nba_128 nba_long_4096; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipelined_leg_$L -- python3 $R/bench.py --only-leg $L --leg-steps 40 > $O/prof_pipelined_leg_$L.log 2>&1 || echo "leg $L prof failed"
done
# (3) counters on the product path's launch (the profiler serialises the launches: one lagged launch at a time), separate passes
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$T -- python3 $R/bench.py $BQ --steps 8 --warmup 4 --no-serial-check > $O/pmc_$T.log 2>&1 || { echo "pmc $T failed"; exit 1; }
done
cd $R
# batch-size sweep (pipelined, as the headline)
for S in 64 128 256 512 1024 2048 4096; do
  timeout -k 10 200 python bench.py $BQ --scenes $S --steps 40 --no-serial-check > $O/sweep_s$S.json 2>/dev/null || echo "sweep $S failed"
done
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04col/final_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('frac_serial_equivalent'), d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline'])
print({k: round(v['mean_us']) for k, v in d['kernels'].items()})
PY

#!/bin/bash
# Round-3 measurement artefacts (run through gpurun from the repo root; raw output under gpurun_out/r03c, summaries are copied into
# profiles/r03 by `python profiles/summarize_pmc.py gpurun_out/r03c profiles/r03` in the authoring container):
#   gpurun --timeout 1100 -- 'bash profiles/collect_r03.sh'      then      gpurun --timeout 900 -- 'bash profiles/collect_r03.sh b'
# PMC counters are collected in their own passes (never combined with sys/hip traces), the program directly after "--".
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03c
mkdir -p $O
if [ "$1" != b ]; then rm -rf $O/prof_* $O/pmc_* $O/tl_*; fi   # (stale passes of earlier calls would be merged into the same directories)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), r['kernel'], round(r['frac'],3), round(r.get('frac_serial_equivalent', 0),3), round(r['launches_in_flight'],2))"; }
if [ "$1" = b ]; then
  # fused launch vs the round-2 pipeline (separate per-agent launches, one chain workgroup per CU), alternating on this box; legs likewise
  SIZES="512 256 1024" bash profiles/exp_r03_fused_ab.sh
  cp gpurun_out/r03b/fused_ab_pipelined.txt $O/fused_ab_pipelined.txt; cp gpurun_out/r03b/fused_ab_serial.txt $O/fused_ab_serial.txt
  for F in 1 0; do
    STTODE_FUSED=$F timeout -k 10 300 python bench.py --no-cpu --no-train --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['configs'].items(): print('fused=$F', k, round(v['value']/1e6,2), 'M traj/s', round(v['ms_per_step'],3), 'ms', (v.get('roofline') or {}).get('kernel'), round((v.get('roofline') or {}).get('frac',0),3))"
  done > $O/legs_fused_vs_unfused.txt
  # batch-size sweep (pipelined, as the headline)
  timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 > /dev/null 2>&1     # (first process after a pause: discarded)
  for S in 128 256 512 1024 2048 4096; do
    timeout -k 10 200 python bench.py --legs none --no-cpu --scenes $S --steps 40 --no-exploratory --no-per-scene --no-sustained > $O/sweep_s$S.json 2>/dev/null || echo "sweep $S failed"
  done
  # block-level trace (diagnostic build): who ran where and when
  F="launches, |steady|roles:|groups:|per-CU|clock|role phases"
  STTODE_HIP_LIB=$R/sttode_amd/lib/variants/lib_trace.so TRACE_NAME=c timeout -k 10 200 python profiles/exp_r03_trace.py 512 30 2>&1 | grep -E "$F" > $O/trace_pipelined_fused.txt
  STTODE_HIP_LIB=$R/sttode_amd/lib/variants/lib_trace.so TRACE_NAME=c timeout -k 10 200 python profiles/exp_r03_trace.py 512 10 serial 2>&1 | grep -E "$F" > $O/trace_serial_fused.txt
  STTODE_BF16X3=1 STTODE_HIP_LIB=$R/sttode_amd/lib/variants/lib_trace.so TRACE_NAME=c timeout -k 10 200 python profiles/exp_r03_trace.py 512 30 2>&1 | grep -E "$F" > $O/trace_pipelined_fused_bf16x3.txt
  timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 > /dev/null 2>&1
  for i in 1 2 3; do
    echo "f32: $(timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 --no-exploratory --no-per-scene --no-sustained 2>/dev/null | line)"
    echo "bf16x3: $(STTODE_BF16X3=1 timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 --no-exploratory --no-per-scene --no-sustained 2>/dev/null | line)"
  done > $O/bf16x3_ab.txt
  timeout -k 10 200 python profiles/exp_r03_bf16x3_probe.py $O/bf16x3_probe.json > /dev/null 2>&1
  timeout -k 10 300 python profiles/exp_per_scene_latency.py > $O/per_scene_latency.txt 2>&1
  exit 0
fi
timeout -k 10 600 python bench.py > $O/final_bench.json 2> $O/final_bench.err || { echo "bench failed"; tail -5 $O/final_bench.err; exit 1; }
cd /tmp
# (1) SERIAL kernel stats, no counters: flop_per_launch / AverageNs / 157.3e12 is the plain per-launch roofline fraction
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial_headline -- python3 $R/bench.py --serial --legs none --no-cpu --no-exploratory --no-per-scene --no-sustained --steps 20 --warmup 3 > $O/prof_serial_headline.log 2>&1 || { echo "serial prof failed"; exit 1; }
for L in ucy_2048 sdd_1024 nba_128 nba_long_4096; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial_leg_$L -- python3 $R/bench.py --only-leg $L --serial --leg-steps 16 > $O/prof_serial_leg_$L.log 2>&1 || echo "serial leg $L prof failed"
done
# (2) PIPELINED kernel trace: stats + union of the launch intervals from the trace's own timestamps (cross-check of the HIP-event union)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pipelined_headline -- python3 $R/bench.py --legs none --no-cpu --no-exploratory --no-per-scene --no-sustained --steps 20 --warmup 3 > $O/prof_pipelined_headline.log 2>&1 || echo "pipelined prof failed"
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tl_pipelined -o tl -- python3 $R/bench.py --legs none --no-cpu --no-exploratory --no-per-scene --no-sustained --steps 20 --warmup 3 --no-serial-check > $O/tl_pipelined.log 2>&1 \
  && python3 $R/profiles/summarize_timeline.py $O/tl_pipelined/tl_results.db traj_chain 36 4 > $O/timeline_pipelined.txt
# (3) counters, serial bench, separate passes
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$T -- python3 $R/bench.py --legs none --no-cpu --no-exploratory --no-per-scene --no-sustained --steps 3 --warmup 1 --serial > $O/pmc_$T.log 2>&1 || { echo "pmc $T failed"; exit 1; }
done
cd $R
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03c/final_bench.json'))
r = d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('frac_serial_equivalent'), d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline'])
print({k: round(v['mean_us']) for k, v in d['kernels'].items()})
PY

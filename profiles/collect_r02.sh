#!/bin/bash
# Round-2 measurement artefacts (run through gpurun from the repo root; raw output under gpurun_out/r02c, summaries are copied
# into profiles/r02 by `python profiles/summarize_pmc.py gpurun_out/r02c profiles/r02` in the authoring container):
#   gpurun --timeout 1100 -- 'bash profiles/collect_r02.sh'
# PMC counters are collected in their own passes (never combined with sys/hip traces), the program directly after "--".
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02c
mkdir -p $O
timeout -k 10 500 python bench.py > $O/final_bench.json 2> $O/final_bench.err || { echo "bench failed"; tail -5 $O/final_bench.err; exit 1; }
cd /tmp
# kernel stats of the headline leg alone (same workload / pipelining as the bench's timed region)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -- python3 $R/bench.py --legs none --no-cpu --steps 20 --warmup 3 > $O/prof_headline.log 2>&1 || { echo "prof failed"; exit 1; }
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_$T -- python3 $R/bench.py --legs none --no-cpu --steps 3 --warmup 1 --serial > $O/pmc_$T.log 2>&1 || { echo "pmc $T failed"; exit 1; }
done
# the three-kernel form on the same box (A/B of the per-trajectory stage), pipelined as in round 1
cd $R
STTODE_CHAIN=0 STTODE_B_STREAMS=1 timeout -k 10 200 python bench.py --legs none --no-cpu --depth 2 > $O/bench_three_kernel_form.json 2>/dev/null
# batch-size sweep (pipelined, as the headline)
for S in 128 256 512 1024 2048 4096; do
  timeout -k 10 200 python bench.py --legs none --no-cpu --scenes $S --steps 12 > $O/sweep_s$S.json 2>/dev/null || echo "sweep $S failed"
done
timeout -k 10 300 python profiles/exp_nba_config5.py > $O/nba_config5.txt 2>&1
timeout -k 10 300 python profiles/exp_per_scene_latency.py > $O/per_scene_latency.txt 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/r02c/final_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline'])
print({k: round(v['mean_us']) for k, v in d['kernels'].items()})
PY

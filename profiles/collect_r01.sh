#!/bin/bash
# Regenerates the round's measurement artefacts on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash profiles/collect_r01.sh'
# then, back in the authoring container:  python profiles/summarize_pmc.py gpurun_out profiles/r01
# PMC counters are collected in their own passes (no sys/hip trace combined with --pmc), program directly after "--".
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || exit 1
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $R/gpurun_out/prof_r01.log 2>&1 || exit 1
for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $P | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$T -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --serial > $R/gpurun_out/pmc_$T.log 2>&1 || { echo "pmc $T failed"; exit 1; }
done
cd $R
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_final.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline'])
print({k: round(v['mean_us']) for k, v in d['kernels'].items()})
PY

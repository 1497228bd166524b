#!/bin/bash
# Round-2 measurement artefacts (run through gpurun from the repo root; raw output under gpurun_out/r02c, summaries are copied
# into profiles/r02 by `python profiles/summarize_pmc.py gpurun_out/r02c profiles/r02` in the authoring container):
#   gpurun --timeout 1100 -- 'bash profiles/collect_r02.sh'      then      gpurun --timeout 900 -- 'bash profiles/collect_r02.sh b'
# PMC counters are collected in their own passes (never combined with sys/hip traces), the program directly after "--".
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02c
mkdir -p $O
if [ "$1" = b ]; then
# single-scene call: GPU latency with the latency forms off / on, kernel timeline of one call
timeout -k 10 200 python profiles/exp_mlp_latency.py 2>&1 | grep -v amdgpu.ids > $O/single_scene_call_latency.txt
(cd /tmp && NAGENTS=32 LAT_ONLY=1 timeout -k 10 200 rocprofv3 --kernel-trace -d $O/prof_scene -o scene -- python3 $R/profiles/exp_mlp_latency.py > $O/prof_scene.log 2>&1) && python profiles/summarize_timeline.py $O/prof_scene/scene_results.db scene_orig,frontend_small 8 > $O/single_scene_timeline.txt
# training step: host split, kernel timeline of one step, bench lines with torch's fused / foreach Adam
timeout -k 10 200 python profiles/exp_train_timeline.py 2>&1 | grep "ms/step" > $O/train_step_split.txt
(cd /tmp && REPS=20 timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof_train -o tr -- python3 $R/profiles/exp_train_timeline.py > $O/prof_train.log 2>&1) && python profiles/summarize_train_timeline.py $O/prof_train/tr_results.db >> $O/train_step_split.txt
timeout -k 10 200 python bench.py --train --no-cpu 2>/dev/null | tail -1 > $O/train_bench_fused_adam.json
timeout -k 10 200 python bench.py --train --no-cpu --train-adam foreach 2>/dev/null | tail -1 > $O/train_bench_foreach_adam.json
timeout -k 10 200 python profiles/exp_train_step.py 2>&1 | grep -v amdgpu.ids > $O/train_step_eth_nba.txt
# counters of the geodesic attention at the config-5 length (ONE group of 4096 x 10): its own pass, program directly after "--"
(cd /tmp && ONLY_CONFIG5=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_attn_l4096 -- python3 $R/profiles/exp_nba_config5.py > $O/pmc_attn_l4096.log 2>&1) || echo "attention pmc pass failed"
bash profiles/exp_legs_chain_vs_latency.sh > $O/legs_chain_vs_three_kernel.txt 2>&1
exit 0
fi
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
  timeout -k 10 200 python bench.py --legs none --no-cpu --scenes $S --steps 40 > $O/sweep_s$S.json 2>/dev/null || echo "sweep $S failed"
done
timeout -k 10 300 python profiles/exp_nba_config5.py > $O/nba_config5.txt 2>&1
timeout -k 10 300 python profiles/exp_per_scene_latency.py > $O/per_scene_latency.txt 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/r02c/final_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline'])
print({k: round(v['mean_us']) for k, v in d['kernels'].items()})
PY

#!/bin/bash
# PMC passes on the serial bench (each counter set in its own pass, no sys/hip trace): per-kernel counters land under gpurun_out/r02/pmc_<tag>_s<S>
# usage (GPU box, repo root): bash profiles/collect_chain_pmc.sh "128 2048"
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/r02
cd /tmp
for S in $1; do
  i=0
  for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/r02/pmc${i}_s$S -- python3 $R/bench.py --legs none --no-cpu --serial --scenes $S --steps 3 --warmup 1 > $R/gpurun_out/r02/pmc${i}_s$S.log 2>&1 || { echo "pmc $i $S failed"; tail -3 $R/gpurun_out/r02/pmc${i}_s$S.log; }
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r02/pmc*_s*')):
    if d.endswith('.log'): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, c in acc.items():
            if 'traj_chain' in k or 'mlp_block0' in k:
                print(d.split('/')[-1], k, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY

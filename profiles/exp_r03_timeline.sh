#!/bin/bash
# Kernel timelines (rocprofv3 --kernel-trace) of the pipelined headline bench: where each call's launches sit relative to the others'.
#   gpurun --timeout 900 -- 'bash profiles/exp_r03_timeline.sh <out-subdir> NAME:ENV=V,ENV=V ...'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$1
shift
mkdir -p $O
timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 > /dev/null 2>&1
cd /tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( for kv in ${envs//,/ }; do export $kv; done
    timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tl_$name -o tl -- python3 $R/bench.py --legs none --no-cpu --steps 16 --warmup 4 ${BENCH_ARGS} > $O/tl_$name.log 2>&1 ) \
    && python3 $R/profiles/summarize_timeline.py $O/tl_$name/tl_results.db traj_chain 40 9 > $O/timeline_$name.txt || exit 1
  tail -1 $O/timeline_$name.txt
done

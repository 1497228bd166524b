#!/bin/bash
# Round 4, tenth GPU call: the multi-rank code path of bench.py on a ONE-rank RCCL group (gather leg, all-reduces), the self-launch path's
# refusal with too few GPUs, the two-process CPU rehearsal, and the full CPU + GPU suites.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04j
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m "not gpu" > $O/cputests.log 2>&1; tail -3 $O/cputests.log   # (informative on this box: the driver runs the CPU suite in the authoring container)
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -1 $O/gputests.log
STTODE_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --legs none --no-cpu --no-train --no-exploratory --no-per-scene > $O/bench_force_dist.json 2> $O/bench_force_dist.err || { tail -20 $O/bench_force_dist.err; exit 1; }
python -c "
import json
d = json.loads(open('$O/bench_force_dist.json').read().strip().splitlines()[-1])
print('one-rank RCCL group: value', round(d['value']/1e6,2), 'rccl_ranks', d['rccl_ranks'], 'gather', d.get('gather'), 'value_incl_gather', round(d.get('value_incl_gather',0)/1e6,2))"
timeout -k 10 120 python bench.py --gpus 2 --steps 2 --warmup 1 --legs none --no-cpu > $O/bench_two_gpus.out 2>&1; echo "bench --gpus 2 on a one-GPU box: exit $?" | tee $O/bench_two_gpus.txt; tail -2 $O/bench_two_gpus.out

#!/bin/bash
# Round 4: the one-scene-per-call loop after removing the scratch array of the role front-end (16 serialised loads at the head of every call).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04s
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "scene or one_launch or latency or golden or fused_launch" > $O/gputests_scene.log 2>&1 || { tail -40 $O/gputests_scene.log; exit 1; }
tail -2 $O/gputests_scene.log
for i in 1 2 3; do timeout -k 10 200 python profiles/exp_per_scene_latency.py 2>/dev/null | tee -a $O/per_scene_latency.txt; done
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_per_scene -- python3 $R/profiles/exp_per_scene_latency.py > $O/prof_per_scene.log 2>&1 || echo "prof failed"
head -5 $O/prof_per_scene/*/*_kernel_stats.csv | cut -c1-150

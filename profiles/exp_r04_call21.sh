#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04u
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "weight_change or golden or per_scene or scene_launch or one_launch or serial" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
for i in 1 2 3; do timeout -k 10 200 python profiles/exp_r03_host_profile.py 2>/dev/null | grep "ms/scene" | tee -a $O/per_scene.txt; done

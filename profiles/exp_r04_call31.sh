#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04ad
mkdir -p $O
for i in 1 2 3; do
echo "epoch flags          : $(timeout -k 10 200 python profiles/exp_r03_host_profile.py 2>/dev/null | grep 'ms/scene')" | tee -a $O/per_scene_epoch_ab.txt
echo "memset per launch    : $(STTODE_SCENE_MEMSET=1 timeout -k 10 200 python profiles/exp_r03_host_profile.py 2>/dev/null | grep 'ms/scene')" | tee -a $O/per_scene_epoch_ab.txt
done

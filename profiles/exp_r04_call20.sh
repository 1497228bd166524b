#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04t
mkdir -p $O
timeout -k 10 200 python profiles/exp_r04_train_shapes.py eth 2>&1 | grep -v amdgpu.ids > $O/train_shapes_eth.txt; head -90 $O/train_shapes_eth.txt

#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04af
mkdir -p $O
timeout -k 10 200 python profiles/exp_r04_tgemm_group_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/tgemm_group_rate.txt

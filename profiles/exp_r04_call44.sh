#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04ai
mkdir -p $O
timeout -k 10 300 python profiles/exp_r04_train_stress.py 2>&1 | grep paired= | tee -a $O/train_stress.txt
STTODE_TRAIN_PAIRED=0 timeout -k 10 300 python profiles/exp_r04_train_stress.py 2>&1 | grep paired= | tee -a $O/train_stress.txt
STTODE_TRAIN_GRAPHS=0 timeout -k 10 300 python profiles/exp_r04_train_stress.py 2>&1 | grep paired= | sed 's/^/eager (no graphs) /' | tee -a $O/train_stress.txt

#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04aj
mkdir -p $O
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "evaluation_loops or dataset" 2>&1 | tail -2
timeout -k 10 300 python profiles/exp_r04_eval_rate.py 2>&1 | grep eval_scenes | tee $O/eval_scenes_rate.txt

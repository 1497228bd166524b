#!/bin/bash
# Round 4: the LDS-tiled training GEMM (csrc/train.hip tgemm_kernel) against the generic kernels -- tests, NBA-size step time A/B, kernel stats;
# the fused integrator stages (config 5's 40 RK4 steps).
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04t
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "tlinear or training or train or grad or sampler or ode or integrator or nba" > $O/gputests_train.log 2>&1 || { tail -40 $O/gputests_train.log; exit 1; }
tail -2 $O/gputests_train.log
for i in 1 2; do
echo "tgemm on : $(timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_ab.txt
echo "tgemm off: $(STTODE_TGEMM=0 timeout -k 10 200 python profiles/exp_train_nba_profile.py 2>/dev/null | tail -1)" | tee -a $O/train_ab.txt
done
timeout -k 10 300 python profiles/exp_r04_ode.py 512 2>&1 | grep "per call" | tee $O/ode_config5.txt
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_nba -- python3 $R/profiles/exp_train_nba_profile.py > $O/prof_train_nba.log 2>&1 || echo "prof failed"
head -9 $O/prof_train_nba/*/*_kernel_stats.csv | cut -c1-150

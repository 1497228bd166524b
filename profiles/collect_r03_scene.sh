#!/bin/bash
# Round 3, item "per-scene drop-in latency": the one-scene-per-call loop (test.py:171-188 pattern) in both forms, the kernel statistics of
# the loop (rocprofv3 --kernel-trace --stats) and the in-launch phase stamps.  Run on the GPU box from the repo root.
O=gpurun_out/r03scene; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 120 python profiles/exp_per_scene_latency.py > $O/one_launch.txt 2>&1
STTODE_SCENE_LAUNCH=0 timeout -k 10 120 python profiles/exp_per_scene_latency.py > $O/six_launch.txt 2>&1
timeout -k 10 120 python profiles/exp_per_scene_latency.py > $O/one_launch_b.txt 2>&1
STTODE_SCENE_LAUNCH=0 timeout -k 10 120 python profiles/exp_per_scene_latency.py > $O/six_launch_b.txt 2>&1
STTODE_HIP_LIB=sttode_amd/lib/variants/lib_sltrace.so timeout -k 10 120 python profiles/exp_r03_scene_trace.py > $O/trace.txt 2>&1
rm -rf $O/prof_one $O/prof_six
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_one -o lat --output-format csv -- python profiles/exp_per_scene_latency.py > $O/prof_one_run.txt 2>&1
STTODE_SCENE_LAUNCH=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/prof_six -o lat --output-format csv -- python profiles/exp_per_scene_latency.py > $O/prof_six_run.txt 2>&1
timeout -k 10 300 python profiles/exp_r03_host_profile.py > $O/host_profile.txt 2>&1
grep "per-scene\|host split" $O/one_launch.txt $O/six_launch.txt $O/one_launch_b.txt $O/six_launch_b.txt

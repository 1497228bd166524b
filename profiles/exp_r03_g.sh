export STTODE_HIP_LIB=$PWD/sttode_amd/lib/variants/lib_trace.so
F="launches, |steady|roles:|groups:|per-CU|clock|role phases"
timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 --no-exploratory > /dev/null 2>&1
echo "== SDD-256 pipelined"; LEG=sdd_1024 TRACE_NAME=sdd timeout -k 10 200 python profiles/exp_r03_trace.py 256 60 2>&1 | grep -E "$F"
echo "== NBA-128 pipelined"; LEG=nba_128 TRACE_NAME=nba timeout -k 10 200 python profiles/exp_r03_trace.py 128 60 2>&1 | grep -E "$F"

export STTODE_HIP_LIB=$PWD/sttode_amd/lib/variants/lib_trace.so
F="steady|roles:|groups:|per-CU|clock|role phases"
timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 --no-exploratory > /dev/null 2>&1
echo "== pipelined fused f32";   TRACE_NAME=f32 timeout -k 10 200 python profiles/exp_r03_trace.py 512 24 2>&1 | grep -E "$F"
echo "== pipelined fused bf16x3";   STTODE_BF16X3=1 TRACE_NAME=b3 timeout -k 10 200 python profiles/exp_r03_trace.py 512 24 2>&1 | grep -E "$F"
echo "== serial fused bf16x3";      STTODE_BF16X3=1 TRACE_NAME=b3 timeout -k 10 200 python profiles/exp_r03_trace.py 512 10 serial 2>&1 | grep -E "$F"

line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline'] or {}; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), round(r.get('frac',0),3), round(d['host_enqueue_ms_per_step'],3))"; }
V=$PWD/sttode_amd/lib/variants
timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 > /dev/null 2>&1
for i in 1 2 3; do
echo "fused prio3: $(timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 2>/dev/null | line)"
echo "fused prio0: $(STTODE_HIP_LIB=$V/lib_prio0.so timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 2>/dev/null | line)"
echo "unfused: $(STTODE_FUSED=0 timeout -k 10 200 python bench.py --legs none --no-cpu --steps 40 2>/dev/null | line)"
done
F="steady|roles:|groups:|per-CU|clock|role phases"
echo "== trace pipelined fused prio3"; STTODE_HIP_LIB=$V/lib_trace.so TRACE_NAME=p3 timeout -k 10 200 python profiles/exp_r03_trace.py 512 30 2>&1 | grep -E "$F"
echo "== trace pipelined fused prio0"; STTODE_HIP_LIB=$V/lib_trace_prio0.so TRACE_NAME=p0 timeout -k 10 200 python profiles/exp_r03_trace.py 512 30 2>&1 | grep -E "$F"

#!/bin/bash
# Round 3: the fused launch (per-agent roles inside the chain launch, two chain workgroups per CU, three launches in flight) against the
# round-2 pipeline (STTODE_FUSED=0: separate per-agent launches, ONE chain workgroup per CU), alternating on one box.
#   gpurun --timeout 1100 -- 'bash profiles/exp_r03_fused_ab.sh'
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03b
mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value']/1e6,2), round(d['ms_per_step'],3), r['kernel'], round(r['frac'],3), round(r['frac_per_launch_latency'],3), round(r['launches_in_flight'],2), {k: round(v['mean_us']) for k, v in d['kernels'].items()})"; }
timeout -k 10 200 python bench.py --legs none --no-cpu --steps 10 > /dev/null 2>&1
for S in ${SIZES:-512 256 1024 128}; do for i in 1 2; do
  for F in 1 0; do
    echo "pipelined scenes=$S fused=$F: $(STTODE_FUSED=$F timeout -k 10 200 python bench.py --legs none --no-cpu --scenes $S --steps 40 2>>$O/err.log | line)"
  done
done; done | tee $O/fused_ab_pipelined.txt
for S in 512 256; do for i in 1 2; do
  for F in 1 0; do
    echo "serial scenes=$S fused=$F: $(STTODE_FUSED=$F timeout -k 10 200 python bench.py --legs none --no-cpu --serial --scenes $S --steps 20 2>>$O/err.log | line)"
  done
done; done | tee $O/fused_ab_serial.txt

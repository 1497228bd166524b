#!/bin/bash
# Round 4, call 24: the groups' predictions through an LDS stage (contiguous 1-KiB stores): parity, headline A/B against the previous build,
# D2H-inclusive rate with the copy on the call's stream vs zero-copy into pinned host memory, stage on / off.
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04x
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not training and not sampler and not tlinear and not pmath" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-serial-check --warmup 5 --steps 20"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), 'incl d2h', round(d['value_incl_d2h']/1e6,2), 'sustained', round(d['sustained']['value']/1e6,2) if 'sustained' in d else None)"; }
V=$PWD/sttode_amd/lib/variants/libsttode_hip_head.so
for i in 1 2; do
echo "stage, D2H on the call's stream : $($B 2>/dev/null | line)" | tee -a $O/ostage_ab.txt
echo "stage, zero-copy predictions    : $(STTODE_BENCH_D2H=zero $B 2>/dev/null | line)" | tee -a $O/ostage_ab.txt
echo "no stage (env), zero-copy       : $(STTODE_CHAIN_OSTAGE=0 STTODE_BENCH_D2H=zero $B 2>/dev/null | line)" | tee -a $O/ostage_ab.txt
echo "previous build, D2H on stream   : $(STTODE_HIP_LIB=$V $B 2>/dev/null | line)" | tee -a $O/ostage_ab.txt
done

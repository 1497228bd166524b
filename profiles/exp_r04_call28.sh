#!/bin/bash
# Round 4, call 28: what the H2D of the step's inputs (1.4 MB, hipMemcpyAsync on the call's stream) costs the pipelined step.
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04ab
mkdir -p $O
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-serial-check --warmup 5 --steps 20"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), 'sustained', round(d['sustained']['value']/1e6,2))"; }
for i in 1 2 3; do
echo "H2D by hipMemcpyAsync (default): $($B 2>/dev/null | line)" | tee -a $O/h2d_ab.txt
echo "H2D by 4 workgroups            : $(STTODE_BENCH_H2D=kernel $B 2>/dev/null | line)" | tee -a $O/h2d_ab.txt
echo "inputs resident (no H2D)       : $(STTODE_BENCH_H2D=resident $B 2>/dev/null | line)" | tee -a $O/h2d_ab.txt
done

#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r04z
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "futures_to_host or zero_copy or lagged or capi" > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 300 python bench.py --no-cpu --no-train --no-exploratory --no-per-scene --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), 'incl d2h', round(d['value_incl_d2h']/1e6,2), 'sustained', round(d['sustained']['value']/1e6,2))"

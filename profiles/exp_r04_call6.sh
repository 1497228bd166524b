#!/bin/bash
# Round 4, sixth GPU call: the driver's command on the current code (clock readings in the line), GPU tests, cold-vs-warm contract region.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r04f
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<PY
import json
d = json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', round(d['value']/1e6,2), 'ms', round(d['ms_per_step'],3), 'clock', d['clock_ghz'], 'frac', round(r['frac'],3), 'path_frac', round(r['path_frac_executed'],3), 'serial frac', r.get('frac_serial_equivalent'))
print('sustained', d['sustained'], 'incl d2h', round(d['value_incl_d2h']/1e6,2))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'parity', d['parity'])
for k, v in d['configs'].items():
    print(k, round(v['value']/1e6,2), round(v['ms_per_step'],3), 'clock', v.get('clock_ghz'), 'frac', round(v['roofline']['frac'],3), 'path', round(v['roofline']['path_frac_executed'],3), 'parity', v.get('parity',{}).get('max_rel_err'))
print('exploratory', {k: v for k, v in d['exploratory_bf16x3'].items() if k in ('value','speedup_vs_f32_headline','parity')})
print('per_scene', d['per_scene']['ms_per_scene'], 'train', d['train']['ms_per_step'], d['train'].get('ms_per_step_foreach_adam'))
PY
B="timeout -k 10 200 python bench.py --legs none --no-cpu --no-train --no-exploratory --no-per-scene --no-serial-check --warmup 5"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), round(d['ms_per_step'],3), 'clock', [round(c,3) for c in d['clock_ghz']], 'sustained', round(d['sustained']['value']/1e6,2) if 'sustained' in d else None)"; }
for i in 1 2 3; do
echo "contract region after the sustained region: $($B --steps 20 2>/dev/null | line)" | tee -a $O/cold_warm.txt
echo "contract region cold (--no-sustained)     : $($B --steps 20 --no-sustained 2>/dev/null | line)" | tee -a $O/cold_warm.txt
done

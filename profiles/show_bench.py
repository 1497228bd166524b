"""Short summary of a bench.py line (the figures DESIGN.md §7 quotes):  python profiles/show_bench.py path/to/final_bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print('headline', round(d['value'] / 1e6, 2), 'M traj/s', round(d['ms_per_step'], 3), 'ms  frac', round(r['frac'], 4), 'path', round(r['path_frac_executed'], 4))
print('sustained', round(d['sustained']['value'] / 1e6, 1), 'incl d2h', round(d['value_incl_d2h'] / 1e6, 1), 'per scene', round(d['per_scene']['ms_per_scene'], 4),
      'bf16x3', round(d['exploratory_bf16x3']['value'] / 1e6, 1))
p = d['parity']
print('parity', p['max_err_over_1_plus_abs_ref'], 'ADE oracle / hip / fused', p['ade_oracle'], p['ade_hip'], p['ade_fused_by_the_call'])
c = d['cpu_baseline']
print('cpu', round(c['value']), 'on', c['cores'], 'threads;', round(c['value_1_thread']), 'on 1;  x', round(d['speedup_vs_cpu_baseline']))
t = d.get('train')
if t:
    print('train one scene', round(t['ms_per_step'], 4), t.get('ms_per_step_quarters'), 'fused', round(t.get('ms_per_step_fused_adam', 0), 4), 'foreach',
          round(t.get('ms_per_step_foreach_adam', 0), 4), 'nba-size', round(t.get('nba_size_step', {}).get('ms_per_step', 0), 4), 'x cpu', round(t.get('speedup_vs_cpu_baseline', 0), 1))
for k, v in d.get('configs', {}).items():
    print(k, round(v['value'] / 1e6, 1), 'M', round(v['ms_per_step'], 3), 'ms path', round(v['roofline']['path_frac_executed'], 3), 'serial',
          round(v['roofline'].get('frac_serial_equivalent') or 0, 2), 'parity', v['parity']['max_err_over_1_plus_abs_ref'])

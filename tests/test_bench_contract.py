"""CPU check of the bench.py output contract on the committed round-1 line (profiles/r01/final_bench.json): the keys the driver
and the judge read must be there with sane values, and the roofline numbers must be self-consistent."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_honours_the_contract():
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r01', 'final_bench.json')))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert 'BASELINE configs[1]' in d['config']['workload'] and d['config']['scenes_per_gpu'] == 512
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert abs(r['achieved'] * 1e12 - r['flop_per_launch'] / r['mean_launch_s']) / (r['achieved'] * 1e12) < 1e-6
    assert 0.3 < r['frac'] < 1.0 and r['traffic'] is not None
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['unit'] == d['unit'] and c['cores'] >= 1
    # throughput and step time describe the same run
    traj = d['config']['trajectories_rank0']
    assert abs(d['value'] - traj / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    assert d['parity']['max_err_over_1_plus_abs_ref'] < 1e-4 and d['parity']['ade_abs_diff'] < 1e-4


def test_committed_round2_line_honours_the_contract():
    """profiles/r02/final_bench.json = the default `python bench.py` line of the round-2 collection run: contract keys, every BASELINE config
    as a leg with parity + roofline + CPU sample, the train object, and a self-consistent busy-time roofline."""
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r02', 'final_bench.json')))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'rccl_ranks', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'configs', 'train', 'value_incl_d2h'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['rccl_ranks'] == 0 and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32'
    assert 'BASELINE configs[1]' in d['config']['workload'] and d['config']['scenes_per_gpu'] == 512 and d['config']['h2d_bytes_per_step'] > 0
    traj = d['config']['trajectories_rank0']
    assert abs(d['value'] - traj / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'flop_per_launch', 'launches', 'busy_time_s', 'mean_launch_s', 'launches_in_flight'):
        assert k in r, k
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.3 < r['frac'] < 1.0
    assert abs(r['achieved'] * 1e12 - r['flop_per_launch'] * r['launches'] / r['busy_time_s']) / (r['achieved'] * 1e12) < 1e-6
    assert abs(r['launches_in_flight'] - r['mean_launch_s'] * r['launches'] / r['busy_time_s']) < 1e-6 and r['launches_in_flight'] >= 0.99
    # the kernel's rate cannot exceed what the step time allows for its share of the work
    assert r['flop_per_launch'] / (d['ms_per_step'] * 1e-3) <= r['achieved'] * 1e12 * 1.05
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['unit'] == d['unit'] and c['cores'] >= 1 and c['value_1_thread'] > 0
    assert d['parity']['max_err_over_1_plus_abs_ref'] < 1e-4 and d['parity']['ade_abs_diff'] < 1e-4
    assert set(d['configs']) == {'ucy_2048', 'sdd_1024', 'nba_128', 'nba_long_4096'}
    for name, leg in d['configs'].items():
        assert leg['value'] > 0 and leg['roofline'] and leg['parity']['max_err_over_1_plus_abs_ref'] < 1e-4, name
        assert leg['cpu_baseline']['value'] > 0 and leg['config']['trajectories_rank0'] > 0
    assert d['train']['steps_per_s'] > 0 and d['train']['cpu_baseline']['value'] > 0


def _run_bench(*argv, env=None):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *argv], capture_output=True, text=True, timeout=240, env=e)


def test_bench_refuses_to_run_fewer_ranks_than_requested():
    """`python bench.py --gpus N` with fewer than N visible GPUs (0 in the CPU container, 1 on the 1-GPU box) must exit non-zero
    and print NO JSON line -- never a single-rank number labelled n_gpus = N (round-1 advisor finding)."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('needs a host with fewer than 2 GPUs')
    r = _run_bench('--gpus', '2', '--steps', '1', '--warmup', '0')
    assert r.returncode != 0
    assert '{' not in r.stdout and 'refusing to run fewer ranks' in r.stderr
    # a launcher-provided world size that disagrees with --gpus is an error too
    r = _run_bench('--gpus', '2', '--steps', '1', '--selftest-dist', env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and '{' not in r.stdout and 'WORLD_SIZE=1 but --gpus 2' in r.stderr


def test_bench_self_launch_spawns_the_ranks():
    """--gpus 2 without a launcher: bench.py starts torch.distributed.run itself (before any GPU call) and the line reports the world
    size the process group saw.  Rehearsed on CPU with gloo (--selftest-dist: same launch / barrier / MAX-over-ranks plumbing)."""
    r = _run_bench('--gpus', '2', '--steps', '3', '--selftest-dist')
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(line) == 1
    d = json.loads(line[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['self_launched'] is True and d['steps'] == 3
    assert d['value'] > 0 and d['backend'].startswith('gloo')
    assert d['gather'] == {'check': 'ok', 'gathered_rows': 5, 'ranks': 2}      # the futures all-gather of the multi-rank line, rehearsed on host rows

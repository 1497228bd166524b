#!/usr/bin/env python3
"""Headline benchmark: predicted-trajectories/sec (K=20 best-of-K) of the STTODE forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path (scene front-end -> MHGSA/ODE encoder -> K=20 decomposition decoder ->
device-side best-of-K ADE/FDE) over one batch of synthetic ETH-shaped scenes that is already resident in HBM.
Workload = BASELINE.json configs[1]: 512 scenes per GPU (<= 32 pedestrians, obs 8 / pred 12, K = 20); weak scaling
(each rank owns its own 512 scenes, no data-path collective; one 3-scalar all-reduce per step aggregates ADE/FDE).

The JSON line carries
  roofline     : dominant kernel's algorithmic FLOP / its mean duration (HIP events on the launch stream, inside the
                 timed region) vs the dense fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md);
  cpu_baseline : the CPU oracle (PyTorch-eager port of the reference path, per-scene loop as test.py:171-184) timed on
                 this box's host cores over a bounded sample of the same scenes (rank 0, N = 1 only);
  parity       : HIP vs oracle on the sampled scenes with injected latents (max relative coordinate error, ADE/FDE).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PEAK_F32_MFMA = 157.3e12  # dense fp32 matrix peak, /opt/skills/guides/MI355X_MICROARCH.md
TP, TF, K = 8, 12, 20

# Algorithmic FLOP per unit (multiply-add = 2), stated in DESIGN.md §4.
F_GRU = TP * (2 * 32 * 6 + 2 * 32 * 288 + 2 * 96 * 288)                        # conv + GRU per column
F_MLP0 = 2 * (2 * 32 * 512 + 2 * 512 * 256) + 2 * 256 * (2 * TP + 2 * TF)        # block-0 x,y MLPs per trajectory
F_MLP1 = 2 * 128 * 512 + 2 * 512 * 256 + 2 * 256 * 2 * TF                       # block-1 y MLP per trajectory
F_LIN = {'A0': 2 * 224 * 512, 'A1': 2 * 128 * 512}
F_ENC = 512 * TP + 8192 * TP + 8192 * TP + 8576 + 24576 + 24576 + 262144 + 256  # per agent, pe part folded (G = 1)
F_TRAJ_SURVEY = 2254214                                                         # SURVEY.md §8d official figure


def kernel_flops(tag, n, m):
    return {'gru_cols[block0,agents]': F_GRU * n, 'gru_cols[block1,trajectories]': F_GRU * m,
            'mlp_block0': F_MLP0 * m, 'mlp_block1': F_MLP1 * m, 'agent_preact': (2 * F_LIN['A0'] + F_LIN['A1']) * n,
            'embed_qkv+post_attn': F_ENC * n}.get(tag)


def train_bench(args, rank, world, dev, dist):
    """Training steps/s: the reference's per-scene loop (train.py:72-95: set_data with augmentation, forward, zero_grad, backward,
    Adam step) over this rank's synthetic ETH scenes; with several ranks the gradients are averaged by one flat all-reduce per
    step (sttode_amd.parallel.average_gradients).  One JSON line, same conventions as the headline bench."""
    from helpers import make_args
    from sttode_amd import STTODENet, parallel, scenes
    from sttode_amd.weights import make_weights, to_torch_state_dict
    sd = to_torch_state_dict(make_weights(1234))
    model = STTODENet(make_args('eth', TP, TF), dev)
    model.load_state_dict(sd, strict=True)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    nsc = 64
    data = [scenes.eth_scene(100000 + rank * nsc + i) for i in range(nsc)]
    data = [(torch.from_numpy(o).to(dev), torch.from_numpy(p).to(dev)) for o, p in data]
    agents = sum(o.shape[0] for o, _ in data) / nsc

    TB = max(1, args.train_batch)
    if TB > 1:                                                       # batched steps: TB consecutive scenes as one CSR batch
        import numpy as _np
        batches = []
        for b0 in range(0, nsc, TB):
            grp = data[b0:b0 + TB]
            ptr = torch.tensor(_np.concatenate([[0], _np.cumsum([o.shape[0] for o, _ in grp])]).astype('int32'), device=dev)
            batches.append((torch.cat([o.permute(0, 2, 1) for o, _ in grp]).contiguous(), torch.cat([p.permute(0, 2, 1) for _, p in grp]).contiguous(), ptr))

    def step(i):
        if TB > 1:
            past, fut, ptr = batches[i % len(batches)]
            model.set_scene_batch(past, fut, ptr)                    # (no augmentation in the batched form)
            o = past
        else:
            o, p = data[i % nsc]
            model.set_data(None, o, p, None, None)
        tot = model.forward()[0]
        opt.zero_grad()
        tot.backward()
        if world > 1:
            parallel.average_gradients(model.parameters(), weight=float(o.shape[0]))
        opt.step()

    for i in range(max(args.warmup, 2 * nsc)):                     # every scene size is seen twice: hipGraphs captured
        step(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    out = {'metric': 'training-steps/sec (one scene per step, forward + backward + Adam)' if TB == 1 else
                     f'training-scenes/sec ({TB} scenes per step, forward + backward + Adam)', 'value': world * args.steps * TB / dt,
           'unit': 'steps/s' if TB == 1 else 'scenes/s', 'n_gpus': args.gpus, 'steps': args.steps, 'warmup': max(args.warmup, 2 * nsc),
           'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
           'data': 'synthetic', 'config': {'workload': f'train.py:72-95 loop over {nsc} synthetic ETH-shaped scenes per GPU (2..32 '
                                                       f'pedestrians, mean {agents:.1f}), obs={TP} pred={TF}, train() mode '
                                                       '(rotation + positional dropout), Adam lr 1e-4',
                                           'parallelism': f'scenes x{world}' + (' + flat gradient all-reduce' if world > 1 else '')}}
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle.sttode_ref import STTODENetRef                 # cpu_baseline leg only
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(16, ncpu)))
        ora = STTODENetRef(make_args('eth', TP, TF)).eval()
        ora.load_state_dict(sd, strict=True)
        oo = torch.optim.Adam(ora.parameters(), lr=1e-4)
        t_cpu, k = 0.0, 0
        while t_cpu < args.cpu_seconds or k < 2:
            o, p = data[k % nsc]
            nn_ = o.shape[0]
            tc = time.perf_counter()
            ora.set_data(None, o.cpu(), p.cpu())
            tot = ora.forward_loss_tensors(torch.randn(nn_, 32), torch.randn(nn_, 32), torch.randn(nn_ * 20, 32))[0]
            oo.zero_grad()
            tot.backward()
            oo.step()
            t_cpu += time.perf_counter() - tc
            k += 1
        out['cpu_baseline'] = {'value': k / t_cpu, 'unit': 'steps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                               'sample': f'{k} steps of the same loop on the PyTorch-eager fp32 oracle (torch autograd), {t_cpu:.1f} s'}
        out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--scenes', type=int, default=512, help='scenes per GPU per step')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the CPU-baseline sample')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--time-every', type=int, default=4, help='bracket the kernels of every n-th step with HIP events (0 = never)')
    ap.add_argument('--serial', action='store_true', help='no cross-step pipelining (one inference() per step)')
    ap.add_argument('--col-parts', type=int, default=0, help='column parts pipelined over streams (0 = library default)')
    ap.add_argument('--train-batch', type=int, default=1, help='with --train: scenes per optimizer step (1 = the reference loop; '
                                                                '>1 = one batched step whose gradient is the sum of the per-scene gradients)')
    ap.add_argument('--train', action='store_true', help='secondary metric: training steps/s (train.py:72-95 loop, one scene per step); '
                                                         'the default run and the headline metric stay the inference path')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    dist = None
    if world > 1 or os.environ.get('STTODE_BENCH_FORCE_DIST'):   # the env switch lets a 1-rank launch exercise the RCCL code path
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    assert world == args.gpus or world == 1, 'launch with torch.distributed.run --nproc-per-node == --gpus'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    from helpers import make_args
    from sttode_amd import STTODENet, capi, scenes
    from sttode_amd.weights import make_weights, to_torch_state_dict

    if args.train:
        return train_bench(args, rank, world, dev, dist)
    model = STTODENet(make_args('eth', TP, TF), dev).eval()
    model.load_state_dict(to_torch_state_dict(make_weights(1234)), strict=True)
    sb = scenes.make_scene_batch(range(rank * args.scenes, (rank + 1) * args.scenes), 'eth')
    n, m = sb.n_agents, sb.n_agents * K
    past, fut = torch.from_numpy(sb.past).to(dev), torch.from_numpy(sb.future).to(dev)
    ptr = torch.from_numpy(sb.scene_ptr).to(dev)
    model.set_scene_batch(past, fut, ptr)
    model.packed()
    if args.col_parts:
        model.native().set_col_parts(args.col_parts)
    n_dev = torch.tensor(float(n), dtype=torch.float32, device=dev)
    acc = None

    # Steps are software-pipelined (depth 2): step i's per-agent stage overlaps step i-1's per-trajectory kernels
    # (sttode_inference_scenes_async); the metrics of step i-1 are taken while step i is in flight.  --serial disables it.
    pending = []

    def finish(h):
        pred = model.wait(h)                               # [K, n, Tf, 2]
        ade, fde = model.best_of_k(pred.permute(1, 0, 2, 3))
        return torch.stack((ade.sum(), fde.sum(), n_dev))   # local sums; ONE 3-scalar all-reduce after the last step (no per-step rank coupling)

    def step():
        # inputs are resident; z is drawn on device by inference() exactly like Normal.rsample in the reference
        model.set_scene_batch(past, fut, ptr)
        if args.serial:
            pred = model.inference(None)
            ade, fde = model.best_of_k(pred.permute(1, 0, 2, 3))
            return torch.stack((ade.sum(), fde.sum(), n_dev))
        pending.append(model.inference_async())
        return finish(pending.pop(0)) if len(pending) > 1 else None

    def drain():
        out = None
        while pending:
            out = finish(pending.pop(0))
        return out

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # per-stage hipEvents recorded on the launch streams by csrc/pipeline.hip, on every 4th step of the timed region
    # (bracketing every step costs 2 % of throughput; measured 67.9 -> 69.2 M traj/s without any brackets)
    model.native().timing(args.time_every)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = step()
        acc = r if r is not None else acc
    r = drain()                                            # every one of the K steps completes inside the timed region
    acc = r if r is not None else acc
    if dist is not None:
        dist.all_reduce(acc)                               # metrics of the last step over all ranks: sum ADE, sum FDE, agents
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stage_ms = model.native().read_timing()
    model.native().timing(0)

    tt = torch.tensor([dt, float(m)], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt, total_traj = float(tmax[0]), float(tt[1])
    else:
        total_traj = float(m)
    value = total_traj * args.steps / dt

    # per-kernel durations from the events recorded inside the timed region (this rank)
    kern, dom = {}, None
    for t, (ms, cnt) in stage_ms.items():
        mean_s = ms * 1e-3 / cnt
        kern[t] = {'mean_us': 1e6 * mean_s, 'launches_sampled': cnt}
        fl = kernel_flops(t, n, m)
        if fl is None:
            continue
        kern[t]['tflops'] = fl / mean_s / 1e12
        if dom is None or ms > dom[1]:
            dom = (t, ms, fl, mean_s)
    roof = None
    if dom:
        traffic = None
        tp = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tp):
            traffic = json.load(open(tp)).get(dom[0])
        roof = {'kernel': dom[0], 'bound': 'mfma', 'achieved': dom[2] / dom[3] / 1e12, 'peak': PEAK_F32_MFMA / 1e12,
                'unit': 'TFLOP/s', 'frac': dom[2] / dom[3] / PEAK_F32_MFMA, 'traffic': traffic,
                'flop_per_launch': dom[2], 'mean_launch_s': dom[3], 'events_every_nth_step': args.time_every,
                'path_frac_executed': value / world * (F_GRU + F_MLP0 + F_MLP1 + (F_ENC + F_GRU + 2 * F_LIN['A0'] + F_LIN['A1']) / K) / PEAK_F32_MFMA,
                'path_frac_survey_flops': value / world * F_TRAJ_SURVEY / PEAK_F32_MFMA}

    out = {'metric': 'predicted-trajectories/sec (20-sample best-of-K)', 'value': value, 'unit': 'trajectories/s',
           'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': f'BASELINE configs[1]: synthetic ETH-shaped scenes (2..32 pedestrians), obs={TP} pred={TF}, '
                                  f'K={K}, {args.scenes} scenes per GPU per step, random-recipe weights (seed 1234)',
                      'scenes_per_gpu': args.scenes, 'agents_rank0': n, 'trajectories_rank0': m, 'parallelism': f'scenes x{world}'},
           'roofline': roof, 'kernels': kern,
           'kernels_note': 'HIP-event durations on the launch streams; in the pipelined run the per-agent stages (frontend, embed_qkv, '
                           'post_attn, gru_cols[block0], agent_preact) execute inside the tails of the previous batch, so their '
                           'durations include waiting for compute units (alone they take 20 / 39 / 44 / 93 / 68 us)'}

    if rank == 0:
        out['ade_fde_synthetic'] = [float(acc[0] / acc[2]), float(acc[1] / acc[2])]
    if rank == 0 and world == 1 and not args.no_cpu:
        from helpers import oracle_model, oracle_scene_inference
        ora = oracle_model('eth', TP, TF)
        # threads: the box's usable cores, but never more than 16 -- the per-scene ops are tiny and PyTorch-CPU gets
        # SLOWER beyond that (256 threads measured 100x slower than 8); the count actually used is reported.
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(16, ncpu)))
        z_all = scenes.latents(99, n)
        # parity sample first (not timed): HIP with injected z vs oracle
        model.set_scene_batch(past, fut, ptr)
        hip = model.inference(None, z=torch.from_numpy(z_all)).cpu().numpy()
        max_rel, traj_cpu, t_cpu, s = 0.0, 0, 0.0, 0
        ade_o, ade_h = [], []
        from oracle.metrics_ref import best_of_k_ade_fde
        while t_cpu < args.cpu_seconds or s < 2:                  # bounded sample: whole passes over the workload's scenes, cycled
            i = s % sb.n_scenes
            a, b = int(sb.scene_ptr[i]), int(sb.scene_ptr[i + 1])
            obs, pr = sb.scene(i)
            tc = time.perf_counter()
            ref = oracle_scene_inference(ora, obs, pr, z_all[a * K:b * K])
            t_cpu += time.perf_counter() - tc
            traj_cpu += (b - a) * K
            if s < sb.n_scenes:                                    # parity of every scene once (first pass)
                err = np.abs(hip[:, a:b] - ref) / (np.abs(ref) + 1.0)
                max_rel = max(max_rel, float(err.max()))
                gt = sb.future[a:b]
                ade_o.append(best_of_k_ade_fde(ref.transpose(1, 0, 2, 3), gt)[0])
                ade_h.append(best_of_k_ade_fde(hip[:, a:b].transpose(1, 0, 2, 3), gt)[0])
            s += 1
        ao, ah = float(np.concatenate(ade_o).mean()), float(np.concatenate(ade_h).mean())
        out['cpu_baseline'] = {'value': traj_cpu / t_cpu, 'unit': 'trajectories/s', 'cores': torch.get_num_threads(), 'host_cpus_visible': ncpu, 'kind': 'port',
                               'sample': f'{s} scene evaluations cycling over the {sb.n_scenes} scenes of this workload, per-scene set_data+inference loop '
                                         f'(test.py:171-184 structure), PyTorch-eager fp32 oracle, {t_cpu:.1f} s of CPU time'}
        out['parity'] = {'scenes_checked': min(s, sb.n_scenes), 'max_err_over_1_plus_abs_ref': max_rel, 'ade_oracle': ao, 'ade_hip': ah,
                         'ade_abs_diff': abs(ao - ah)}
        out['speedup_vs_cpu_baseline'] = value / out['cpu_baseline']['value']
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

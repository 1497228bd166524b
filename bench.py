#!/usr/bin/env python3
"""Headline benchmark: predicted-trajectories/sec (K=20 best-of-K) of the STTODE forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`` (one rank
per GPU, RCCL), or -- when WORLD_SIZE is not in the environment -- bench.py starts exactly that launcher itself as a child
process BEFORE anything touches the GPU and exits with its status.  It never runs fewer ranks than ``--gpus`` asks for: fewer
visible GPUs than N is an error (non-zero exit, no JSON line).  ``n_gpus`` in the line is the world size the process group saw.

A step = one pass of the hot path (H2D of the scene batch from pinned host memory -> scene front-end -> MHGSA/ODE encoder ->
K=20 decomposition decoder -> device-side best-of-K ADE/FDE) over one batch of synthetic scenes (SURVEY.md §8d metric).
Headline workload = BASELINE.json configs[1]: 512 ETH-shaped scenes per GPU (<= 32 pedestrians, obs 8 / pred 12, K = 20);
weak scaling (each rank owns its own scenes, no data-path collective; one 3-scalar all-reduce of ADE/FDE sums at the end).

The JSON line carries
  roofline     : dominant kernel's algorithmic FLOP / its mean duration (HIP events on the launch stream, inside the
                 timed region) vs the dense fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md);
  cpu_baseline : the CPU oracle (PyTorch-eager port of the reference path, per-scene loop as test.py:171-184) timed on
                 this box's host cores over a bounded sample of the same scenes (rank 0, N = 1 only), 16 threads and 1 thread;
  parity       : HIP vs oracle on the sampled scenes with injected latents (max relative coordinate error, ADE/FDE);
  configs      : secondary legs for BASELINE configs 2-5 at the per-GPU share each config implies (UCY-mixed 2048/8 = 256 scenes,
                 SDD 1024/4 = 256 scenes, NBA B=128 x 11 agents (test.py:616-622), NBA long horizon 4096/8 = 512 scenes x 10 agents,
                 obs 10 / pred 40), each with ms_per_step, trajectories/s, its dominant kernel's roofline and a short CPU sample;
  sustained    : the headline workload, the same step, as an 80-step run (pipeline fill and drain weigh 1/4 of the 20-step run's)
                 -- a second figure, never `value`;
  per_scene    : the reference's evaluation call pattern, ONE scene per call (set_data + inference + .cpu(), test.py:171-188): ms per scene;
  train        : training steps/s of the train.py loop, with its own CPU baseline.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

PEAK_F32_MFMA = 157.3e12  # dense fp32 matrix peak, /opt/skills/guides/MI355X_MICROARCH.md
K = 20
F_TRAJ_SURVEY = 2254214   # SURVEY.md §8d figure for ETH shapes, written before the per-agent / per-trajectory layer-1 split


# Algorithmic FLOP per unit (multiply-add = 2), stated in DESIGN.md §4.
def flops(Tp, Tf, G=1):
    f = {'gru': Tp * (2 * 32 * 6 + 2 * 32 * 288 + 2 * 96 * 288),                                # conv + GRU per column
         'mlp0': 2 * (2 * 32 * 512 + 2 * 512 * 256) + 2 * 256 * (2 * Tp + 2 * Tf),               # block-0 x,y MLPs per trajectory
         'mlp1': 2 * 128 * 512 + 2 * 512 * 256 + 2 * 256 * 2 * Tf,                               # block-1 y MLP per trajectory
         'A0': 2 * 224 * 512, 'A1': 2 * 128 * 512,
         'enc': 512 * Tp + 8192 * Tp + 8192 * Tp + 8576 + 24576 + 24576 + 262144 + 256 * G,     # per agent, pe part folded
         'attn': 4 * G * 64}                                                                      # scores + PV per agent (VALU)
    f['path_per_traj'] = f['gru'] + f['mlp0'] + f['mlp1'] + (f['enc'] + f['gru'] + 2 * f['A0'] + f['A1']) / K
    return f


def kernel_flops(tag, n, m, F):
    return {'gru_cols[block0,agents]': F['gru'] * n, 'gru_cols[block1,trajectories]': F['gru'] * m,
            'mlp_block0': F['mlp0'] * m, 'mlp_block1': F['mlp1'] * m, 'agent_preact': (2 * F['A0'] + F['A1']) * n,
            'trajectory_chain': (F['mlp0'] + F['gru'] + F['mlp1']) * m,
            # the fused launch (round 3) also runs the per-agent stage: encoder, block-0 GRU, the three layer-1 tables
            'agents+trajectory_chain[fused launch]': (F['mlp0'] + F['gru'] + F['mlp1']) * m + (F['enc'] + F['gru'] + 2 * F['A0'] + F['A1']) * n}.get(tag)


# ------------------------------------------------------------------------------------------------------------------
# launch: self-spawn of the ranks (no GPU call in this process), or failure
# ------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """--gpus N > 1 without a launcher: start ``torch.distributed.run`` with N ranks as a child and exit with its status.
    Nothing in this process has touched the GPU (torch.cuda.device_count() does not initialise HIP on this image)."""
    import torch
    backend_cpu = bool(args.selftest_dist) or args.dist_backend == 'gloo'   # (gloo rehearsal: the ranks share cuda:0)
    have = torch.cuda.device_count()
    if not backend_cpu and have < args.gpus:
        sys.stderr.write(f'bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible; refusing to run fewer ranks '
                         f'than requested (no JSON line is printed)\n')
        return 2
    port = 29400 + os.getpid() % 500
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), STTODE_BENCH_SELF_LAUNCHED='1')
    return subprocess.run(cmd, env=env).returncode


def init_dist(args):
    """-> (rank, world, local, dist | None).  A WORLD_SIZE that disagrees with --gpus is an error, never a relabelled run."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        sys.stderr.write(f'bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch with --nproc-per-node == --gpus\n')
        sys.exit(2)
    dist = None
    if world > 1 or os.environ.get('STTODE_BENCH_FORCE_DIST'):   # the env switch lets a 1-rank launch exercise the RCCL code path
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.selftest_dist:
            dist.init_process_group('gloo')
        elif args.dist_backend == 'gloo':
            # REHEARSAL (hidden flag, tests/test_gpu_parity.py): every rank computes on cuda:0, the collectives run under gloo on host copies --
            # every multi-rank branch of this file (legs, train with average_gradients, the gather leg) executes with HIP kernels on a
            # one-GPU box.  Its numbers are not a scaling measurement.
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        if dist.get_world_size() != args.gpus:
            sys.stderr.write(f'bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}\n')
            sys.exit(2)
    return rank, world, local, dist


def selftest_dist(args):
    """CPU rehearsal of the launch / timing / reduction plumbing (gloo, no GPU, no kernels): every rank 'processes' a fixed number of
    units per step; the line it prints has the same launch-related keys as the real one.  Used by tests/test_bench_contract.py."""
    import torch
    rank, world, local, dist = init_dist(args)
    units = 1000 * (rank + 1)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(units)], dtype=torch.float64)
    total = float(units)
    if dist is not None:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt, total = float(tmax[0]), float(tt[1])
    gather = None
    if dist is not None and not args.no_gather_futures:            # the collective of gather_futures_leg on ragged host rows (gloo)
        from sttode_amd import parallel
        rows = torch.full((rank + 2, 4), float(rank))
        allr = parallel.gather_futures(rows)
        want_rows = sum(r + 2 for r in range(world))
        gather = {'check': 'ok' if allr.shape[0] == want_rows and float(allr.sum()) == float(sum(4 * r * (r + 2) for r in range(world))) else 'MISMATCH',
                  'gathered_rows': int(allr.shape[0]), 'ranks': world}
    if rank == 0:
        print(json.dumps({'metric': 'selftest units/sec', 'value': total * args.steps / dt, 'n_gpus': world, 'gather': gather,
                          'rccl_ranks': dist.get_world_size() if dist is not None else 0, 'backend': 'gloo (selftest, no GPU)',
                          'self_launched': bool(os.environ.get('STTODE_BENCH_SELF_LAUNCHED')), 'steps': args.steps}))
    if dist is not None:
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------------------------
# one workload ("leg"): inputs in pinned host memory, two device input slots, pipelined steps
# ------------------------------------------------------------------------------------------------------------------
LEGS = {
    # name: (kind, dataset, Tp, Tf, per-GPU size, description)
    'eth_512': ('scenes', 'eth', 8, 12, 512, 'BASELINE configs[1]: synthetic ETH-shaped scenes (2..32 pedestrians)'),
    'ucy_2048': ('scenes', 'ucy', 8, 12, 256, 'BASELINE configs[2]: UCY-mixed (zara1/zara2 2..20, univ 20..60 pedestrians), 2048 scenes / 8 GPUs'),
    'sdd_1024': ('scenes', 'sdd', 8, 12, 256, 'BASELINE configs[3]: SDD ragged scenes (1..40 agents, pixels/50), 1024 scenes / 4 GPUs'),
    'nba_128': ('nba', 'nba', 5, 10, 128, 'reference NBA test batch: one attention group of B=128 scenes x 11 agents (test.py:616-622)'),
    'nba_long_4096': ('nba', 'nba', 10, 40, 512, 'BASELINE configs[4]: NBA long horizon, 10 agents, obs 10 / pred 40, 4096 scenes / 8 GPUs = one '
                                                  'attention group of 512 scenes per GPU (groups -> ranks, SURVEY.md §8e)'),
}


def allreduce(dist, t, op=None):
    """dist.all_reduce of a device tensor under whatever backend the group runs on (gloo rehearsal: through a host copy)."""
    import torch.distributed as td
    op = td.ReduceOp.SUM if op is None else op
    if dist.get_backend() == 'gloo' and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


class Leg:
    # futures to the host (value_incl_d2h) = STTODENet.futures_to_host_async: a few persistent workgroups copy them to pinned memory on the
    # call's own stream (75-76 M traj/s beside a 77 M headline; the alternatives measured in round 4 -- hipMemcpyAsync on the call's stream 63 M,
    # on a dedicated stream 55 M, zero-copy 13-45 M -- are in profiles/r04/d2h_placement_ab.txt and profiles/exp_d2h_placement.py, not here)
    STREAMS = 3      # pipeline streams the lagged calls rotate over (the library default); calls in flight = 2 x STREAMS slots

    def __init__(self, name, rank, dev, size=None):
        import torch
        from helpers import make_args
        from sttode_amd import STTODENet, scenes
        from sttode_amd.weights import make_weights, to_torch_state_dict
        self.name = name
        self.kind, self.dataset, self.Tp, self.Tf, dsize, self.desc = LEGS[name]
        self.size = size or dsize
        self.dev = dev
        self.model = STTODENet(make_args('nba' if self.kind == 'nba' else 'eth', self.Tp, self.Tf), dev).eval()
        self.model.load_state_dict(to_torch_state_dict(make_weights(1234, past_length=self.Tp, future_length=self.Tf)), strict=True)
        if self.kind == 'scenes':
            base = {'eth': 0, 'ucy': 0, 'sdd': 0}[self.dataset]
            self.sb = scenes.make_scene_batch(range(base + rank * self.size, base + (rank + 1) * self.size), self.dataset)
            self.n = self.sb.n_agents
            host = [torch.from_numpy(self.sb.past), torch.from_numpy(self.sb.future), torch.from_numpy(self.sb.scene_ptr)]
            self.G = 1
        else:
            self.N = 11 if name == 'nba_128' else 10
            self.batch = scenes.nba_batch(7000 + rank, self.size, N=self.N, obs_len=self.Tp, pred_len=self.Tf)
            self.n = self.size * self.N
            host = [torch.from_numpy(self.batch['past_traj']), torch.from_numpy(self.batch['future_traj'])]
            self.G = self.size
        self.m = self.n * K
        self.F = flops(self.Tp, self.Tf, self.G)
        # the step's inputs travel as ONE pinned buffer -> ONE device buffer (one copy per step; the tensors are 256-byte aligned views):
        # every extra small operation on the caller's stream waits for workgroup slots of a chip that the chain launches keep full
        offs, tot = [], 0
        for t in host:
            offs.append(tot)
            tot += (t.numel() * t.element_size() + 255) // 256 * 256
        self.host_buf = torch.empty(tot, dtype=torch.uint8).pin_memory()

        def views(buf):
            return [buf[o:o + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for o, t in zip(offs, host)]
        for v, t in zip(views(self.host_buf), host):
            v.copy_(t)
        self.host = views(self.host_buf)
        self.h2d_bytes = sum(t.numel() * t.element_size() for t in self.host)
        self.depth = 2 * Leg.STREAMS                              # calls in flight (= model.async_depth): device input / workspace slots
        self.model.async_depth = self.depth
        self.slot_bufs = [torch.empty(tot, dtype=torch.uint8, device=dev) for _ in range(self.depth)]
        self.slots = [views(b) for b in self.slot_bufs]
        self.n_dev = torch.tensor(float(self.n), dtype=torch.float32, device=dev)
        self.calls = 0
        self.gather_counts = None
        self.pending = []
        self.d2h_bufs = None
        self.model.packed()

    def _load(self):
        """H2D of this step's inputs (pinned -> one of two device slots, on the caller's stream) + the data-entry call."""
        slot = self.slots[self.calls % self.depth]
        self.slot_bufs[self.calls % self.depth].copy_(self.host_buf, non_blocking=True)
        self.calls += 1
        if self.kind == 'scenes':
            self.model.set_scene_batch(slot[0], slot[1], slot[2])
        else:
            self.model.set_data_nba({'past_traj': slot[0], 'future_traj': slot[1]})

    def _finish(self, h):
        # best-of-K on the call's own pipeline stream, behind the launch that carries its trajectory groups (stream order, no event).
        # Nothing goes onto the caller's stream here; whoever needs the call's outputs or its slot waits for the event (settle()).
        import torch
        self.last_pred = h['pred']
        self.unsettled = h
        out = self.model.best_of_k_async(h, gt=h['gt'])        # per-agent (ade, fde) of the slot; summed ONCE, after the last step
        if self.d2h_bufs is not None:                           # by a few persistent workgroups on ITS stream, behind its groups
            self.model.futures_to_host_async(h, out=self.d2h_bufs[h['slot'] % len(self.d2h_bufs)], workgroups=8)
        if self.gather_counts is not None:
            self._gather(h)
        return out

    def _gather(self, h):
        """All-gather of a finished call's futures on the COMMUNICATION stream: it waits for the call's event (sttode_wait: no kernel), later
        calls keep launching under it; `gather_counts` were exchanged once before the region -- no count collective, no .tolist() sync per step.
        The slot's next user waits for this gather's event before its launch may overwrite the futures (step())."""
        import torch
        from sttode_amd import parallel
        with torch.cuda.stream(self.comm_stream):
            self.model.wait(h)
            self.gathered = parallel.gather_futures(h['pred'], counts=self.gather_counts, reuse=True)
            ev = self.gather_events.get(h['slot'])
            if ev is None:
                ev = self.gather_events[h['slot']] = torch.cuda.Event()
            ev.record(self.comm_stream)

    def settle(self):
        """(an event wait on the caller's stream, no kernel) the latest finished call's launch and metrics are complete: its futures may be
        copied, its metric values read, and the input slot it used may be overwritten."""
        h, self.unsettled = getattr(self, 'unsettled', None), None
        if h is not None:
            self.model.wait(h)

    def sums(self, af):
        import torch
        self.settle()
        return torch.stack((af[0].sum(), af[1].sum(), self.n_dev))   # local sums; ONE all-reduce after the last step

    def step(self, serial=False):
        """ONE step form for the headline and every leg.  Pipelined (default): everything of call k is enqueued on ITS pipeline stream, in
        order -- H2D of the inputs, the latents, the launch (its per-agent roles + the trajectory groups of the call made STREAMS calls
        earlier on that stream), then the metrics of that earlier call -- so there is no cross-stream event anywhere, and no stream carries
        a chain of small kernels that every step has to wait for (round 3's default step kept H2D, latents and metrics on the caller's
        stream: starved by the chain launches they take ~1 ms each there and became the critical path in long runs,
        profiles/r04/cadence_default_80.txt)."""
        import torch
        if serial:
            self._load()
            pred = self.model.inference(None)
            self.last_pred = self.model.diverse_pred            # contiguous [n, K, Tf, 2]
            return self.model.best_of_k(pred.permute(1, 0, 2, 3))
        st = self.model.next_async_stream(self.n)
        if st is None:
            raise RuntimeError('bench.py: the workload does not take the pipelined chain form')
        with torch.cuda.stream(st):
            if self.gather_counts is not None:                  # the slot this call takes: its previous futures have been gathered
                ev = self.gather_events.get(self.model._async_calls % self.depth)
                if ev is not None:
                    st.wait_event(ev)
            self._load()
            # latents z ~ N(0, I) like Normal.rsample in the reference, drawn by the call's own launch; best-of-K ADE / FDE against the
            # batch's futures computed by the call's trajectory groups
            h = self.model.inference_async(metrics_gt=self.model._future)
        h['gt'] = self.model._future
        self.pending.append(h)
        return self._finish(self.pending.pop(0)) if len(self.pending) > Leg.STREAMS else None

    def drain(self):
        out = None
        while self.pending:
            out = self._finish(self.pending.pop(0))
        return out

    def timed(self, steps, warmup, dist, time_every, serial=False, d2h=False, gather=False, prewarm=0):
        """W untimed + exactly `steps` timed steps, barrier + synchronize on both sides, MAX over ranks.
        d2h: additionally copy every step's futures to pinned host memory inside the timed region.
        gather: additionally all-gather every step's futures [n_r, K, Tf, 2] over the ranks (parallel.gather_futures: RCCL over xGMI),
        which is what the reference's metric path needs when the futures are wanted on one rank (test.py:194,526)."""
        import torch
        from sttode_amd import capi, parallel
        dev = self.dev
        hostbuf = torch.empty((self.n, K, self.Tf, 2), dtype=torch.float32).pin_memory() if d2h else None   # contiguous D2H targets
        self.d2h_bufs = [hostbuf] + [torch.empty_like(hostbuf).pin_memory() for _ in range(Leg.STREAMS - 1)] if d2h and not serial else None   # (one per pipeline stream)
        self.gather_counts = None
        if gather and dist is not None:
            # every rank's row count, exchanged ONCE in front of the region (weak scaling: the shards are fixed for the whole run)
            cnt = torch.zeros(dist.get_world_size(), dtype=torch.int64)
            cnt[dist.get_rank()] = self.n
            cnt = cnt.to(dev) if dist.get_backend() != 'gloo' else cnt
            dist.all_reduce(cnt)
            counts = [int(c) for c in cnt.tolist()]
            if not hasattr(self, 'comm_stream'):
                self.comm_stream, self.gather_events = torch.cuda.Stream(), {}
            if not serial:
                self.gather_counts = counts
        acc = None
        import gc
        gc.collect()                                              # before the warm-up: a collector run between warm-up and region would idle the GPU
        gc.disable()                                              # no collector pause inside a timed region of a few milliseconds
        clk = torch.zeros(4, dtype=torch.int64, device=dev)
        for _ in range(prewarm + warmup):                         # (prewarm: extra untimed steps in front of the W warm-up steps -- the shader clock
            self.step(serial)                                     # needs 25-40 ms of load to climb from ~2.1 to 2.4 GHz after an idle gap; --prewarm)
        self.drain()
        capi.call('sttode_clock_probe', clk, capi.stream_ptr())   # shader clock the region starts with (20 us, ahead of the synchronize)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        self.model.native().timing(time_every)
        t0 = time.perf_counter()
        for _ in range(steps):
            r = self.step(serial)
            if r is not None:
                acc = r
                if d2h and serial:
                    self.settle()                                 # the futures about to be copied are complete
                    hostbuf.copy_(self.last_pred, non_blocking=True)
                if gather and serial and dist is not None:
                    self.gathered = parallel.gather_futures(self.last_pred, counts=counts, reuse=True)
        t_host = time.perf_counter() - t0                         # the host's share: enqueueing `steps` steps (no device sync inside)
        r = self.drain()                                          # every one of the K steps completes inside the timed region
        if r is not None:
            acc = r
            if d2h and serial:
                self.settle()
                hostbuf.copy_(self.last_pred, non_blocking=True)
        if self.gather_counts is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)   # the last gathers complete inside the timed region
        self.d2h_bufs, self.gather_counts = None, None
        acc = self.sums(acc)                                      # the last step's per-agent best-of-K values -> (sum ADE, sum FDE, agents)
        if dist is not None:
            allreduce(dist, acc)                                  # metrics of the last step over all ranks
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        capi.call('sttode_clock_probe', clk[2:], capi.stream_ptr())   # ... and the clock it ends with (outside the region)
        c = clk.tolist()
        clock = [c[0] / (10.0 * c[1]) if c[1] else None, c[2] / (10.0 * c[3]) if c[3] else None]
        stage_ms = self.model.native().read_timing()
        self.busy_ms = dict(getattr(self.model.native(), 'busy_ms', {}))
        self.model.native().timing(0)
        tt = torch.tensor([dt, float(self.m)], dtype=torch.float64, device=dev)
        total = float(self.m)
        if dist is not None:
            tmax = tt.clone()
            allreduce(dist, tmax, op=dist.ReduceOp.MAX)
            allreduce(dist, tt, op=dist.ReduceOp.SUM)
            dt, total = float(tmax[0]), float(tt[1])
        return {'dt': dt, 'total_traj': total, 'value': total * steps / dt, 'ms_per_step': 1e3 * dt / steps, 'stage_ms': stage_ms,
                'metrics': acc, 'host_ms_per_step': 1e3 * t_host / steps, 'clock_ghz': clock}

    def roofline(self, stage_ms, value_per_gpu, time_every, calls=None):
        """Dominant kernel's rate.  Launches of consecutive pipelined steps run CONCURRENTLY (three streams), so the rate is
        (algorithmic FLOP of the region's calls) / busy time,  busy time = length of the union of the launches' [start, end] HIP-event
        intervals; FLOP / mean launch duration would count the shared time once per launch in flight.  The raw mean launch duration (what
        rocprofv3 --stats averages) and the mean number of launches in flight are reported next to it.
        Lagged launches: a launch carries the per-agent roles of one call and the trajectory groups of another, and a region of `calls`
        steps makes calls + STREAMS launches (its first STREAMS launches carry roles only, its last STREAMS groups only): the FLOP are
        counted per CALL -- every call's roles and groups run exactly once inside the region -- not per launch."""
        kern, dom = {}, None
        for t, (ms, cnt) in stage_ms.items():
            mean_s = ms * 1e-3 / cnt
            busy_s = max(self.busy_ms.get(t, ms), 1e-9) * 1e-3
            kern[t] = {'mean_us': 1e6 * mean_s, 'launches_sampled': cnt, 'busy_ms': 1e3 * busy_s}
            fl = kernel_flops(t, self.n, self.m, self.F)
            if fl is None:
                continue
            fl_call = fl
            if calls is not None and t == 'agents+trajectory_chain[fused launch]' and cnt > calls:
                kern[t]['launches_sampled'] = cnt
                kern[t]['calls'] = calls
                fl = fl * calls / cnt                               # mean FLOP per launch of this region
            kern[t]['tflops'] = fl * cnt / busy_s / 1e12
            if dom is None or ms > dom[1]:
                dom = (t, ms, fl, mean_s, cnt, busy_s, fl_call)
        roof = None
        if dom:
            traffic = None
            tp = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tp):
                traffic = json.load(open(tp)).get(self.name, {}).get(dom[0])
            ach = dom[2] * dom[4] / dom[5]
            roof = {'kernel': dom[0], 'bound': 'mfma', 'achieved': ach / 1e12, 'peak': PEAK_F32_MFMA / 1e12,
                    'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA, 'traffic': traffic,
                    'flop_per_launch': dom[2], 'flop_per_call': dom[6], 'launches': dom[4], 'busy_time_s': dom[5], 'mean_launch_s': dom[3],
                    'launches_in_flight': dom[3] * dom[4] / dom[5],
                    'achieved_definition': 'flop_per_launch * launches / busy_time_s (union of the launch intervals); with launches_in_flight = 1 '
                                           'this is flop_per_launch / mean_launch_s',
                    'frac_per_launch_latency': dom[2] / dom[3] / PEAK_F32_MFMA,
                    'events_every_nth_step': time_every,
                    # whole-path figure: executed (de-duplicated, layer-1 split) FLOP per trajectory x throughput / peak
                    'path_frac_executed': value_per_gpu * self.F['path_per_traj'] / PEAK_F32_MFMA,
                    'path_flop_per_trajectory': self.F['path_per_traj']}
        return roof, kern

    def timed_form_call(self, set_inputs, n):
        """ONE call through the form the timed regions run -- inference_async on its pipeline stream: lagged launch, throughput-form roles,
        latents drawn by the launch itself (Philox), fused metrics -- and what the parity sample needs of it: predictions [K, n, Tf, 2], the
        latents the call REPORTS (fed to the oracle), the per-agent ADE its own groups computed, and the form's name."""
        import torch
        from sttode_amd import capi
        m = self.model
        nat = m.native()
        lagged = bool(capi.lib().sttode_async_is_lagged(nat.h, n))
        st = m.next_async_stream(n)
        with torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream()):
            set_inputs()
            h = m.inference_async(metrics_gt=m._future)
        ade, _ = m.best_of_k_async(h, gt=m._future)
        pred = m.wait(h)
        torch.cuda.synchronize()
        out = pred.cpu().numpy(), h['z'].cpu().numpy(), ade.cpu().numpy(), ('pipelined: lagged launch, throughput-form roles, in-launch Philox latents, '
                                                                             'fused metrics' if lagged else 'pipelined (round-3 form: below the chain threshold)')
        m.reset_async()
        return out

    # ---- CPU oracle sample + parity (rank 0, N = 1 only; outside every timed region) ----
    def cpu_sample(self, seconds, threads):
        import torch
        from helpers import oracle_model, oracle_scene_inference
        from oracle.metrics_ref import best_of_k_ade_fde
        from sttode_amd import scenes
        torch.set_num_threads(threads)
        ora = oracle_model('nba' if self.kind == 'nba' else 'eth', self.Tp, self.Tf)
        out = {'cores': torch.get_num_threads(), 'unit': 'trajectories/s', 'kind': 'port'}
        if self.kind == 'scenes':
            sb = self.sb
            # the TIMED form is what is held to the oracle (round-4 review: the sample used to go through the serial inference()): the oracle
            # is fed the latents the call reports
            dv = [torch.from_numpy(a).to(self.dev) for a in (sb.past, sb.future, sb.scene_ptr)]
            hip, z_all, ade_dev, form = self.timed_form_call(lambda: self.model.set_scene_batch(*dv), self.n)
            max_rel, traj_cpu, t_cpu, s = 0.0, 0, 0.0, 0
            ade_o, ade_h, ade_d = [], [], []
            while t_cpu < seconds or s < 2:                       # bounded sample: scenes of this workload, cycled
                i = s % sb.n_scenes
                a, b = int(sb.scene_ptr[i]), int(sb.scene_ptr[i + 1])
                obs, pr = sb.scene(i)
                tc = time.perf_counter()
                ref = oracle_scene_inference(ora, obs, pr, z_all[a * K:b * K])
                t_cpu += time.perf_counter() - tc
                traj_cpu += (b - a) * K
                if s < sb.n_scenes:                                # parity of every sampled scene once
                    err = np.abs(hip[:, a:b] - ref) / (np.abs(ref) + 1.0)
                    max_rel = max(max_rel, float(err.max()))
                    gt = sb.future[a:b]
                    ade_o.append(best_of_k_ade_fde(ref.transpose(1, 0, 2, 3), gt)[0])
                    ade_h.append(best_of_k_ade_fde(hip[:, a:b].transpose(1, 0, 2, 3), gt)[0])
                    ade_d.append(ade_dev[a:b])
                s += 1
            ao, ah = float(np.concatenate(ade_o).mean()), float(np.concatenate(ade_h).mean())
            out.update(value=traj_cpu / t_cpu, sample=f'{s} scene evaluations cycling over the {sb.n_scenes} scenes of this workload, per-scene '
                       f'set_data+inference loop (test.py:171-184 structure), PyTorch-eager fp32 oracle, {t_cpu:.1f} s of CPU time')
            par = {'scenes_checked': min(s, sb.n_scenes), 'max_err_over_1_plus_abs_ref': max_rel, 'ade_oracle': ao, 'ade_hip': ah,
                   'ade_abs_diff': abs(ao - ah), 'ade_fused_by_the_call': float(np.concatenate(ade_d).mean()), 'form_checked': form,
                   'latents': 'drawn by the call (reported z fed to the oracle)'}
            return out, par
        # NBA: one forward call on a reduced batch (the attention group is the batch, so HIP runs the SAME reduced batch for parity)
        Bs = min(self.size, 16 if self.Tf > 12 else 32)
        d = {k: (v[:Bs] if isinstance(v, np.ndarray) else v) for k, v in self.batch.items()}
        nn_ = Bs * self.N
        dd = {k: (torch.from_numpy(v).to(self.dev) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
        nat = self.model.native()
        nat.set_chain(1)                                           # the reduced batch through the chain launch the full batch takes
        try:
            hip, z, ade_dev, form = self.timed_form_call(lambda: self.model.set_data_nba(dd), nn_)
        finally:
            nat.set_chain(-1)
        t_cpu, reps = 0.0, 0
        while t_cpu < seconds or reps < 1:
            tc = time.perf_counter()
            with torch.no_grad():
                dt_ = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
                ora.set_data_nba(dt_)
                ref = ora.inference(dt_, z=torch.from_numpy(z)).numpy()
            t_cpu += time.perf_counter() - tc
            reps += 1
        err = np.abs(hip - ref) / (np.abs(ref) + 1.0)
        gt = d['future_traj'].reshape(nn_, self.Tf, 2)
        ao = float(best_of_k_ade_fde(ref.transpose(1, 0, 2, 3), gt)[0].mean())
        ah = float(best_of_k_ade_fde(hip.transpose(1, 0, 2, 3), gt)[0].mean())
        out.update(value=reps * nn_ * K / t_cpu, sample=f'{reps} forward call(s) of a reduced batch (B={Bs} of {self.size} scenes x {self.N} agents: the attention '
                   f'group is the batch, so the CPU cost per trajectory is a lower bound for the full group), PyTorch-eager fp32 oracle, {t_cpu:.1f} s')
        par = {'batch_checked': Bs, 'max_err_over_1_plus_abs_ref': float(err.max()), 'ade_oracle': ao, 'ade_hip': ah, 'ade_abs_diff': abs(ao - ah),
               'ade_fused_by_the_call': float(ade_dev.mean()), 'form_checked': form, 'latents': 'drawn by the call (reported z fed to the oracle)'}
        return out, par

    def config(self, world):
        c = {'workload': f'{self.desc}, obs={self.Tp} pred={self.Tf}, K={K}, {self.size} scenes per GPU per step, random-recipe weights (seed 1234)',
             'scenes_per_gpu': self.size, 'agents_rank0': self.n, 'trajectories_rank0': self.m, 'h2d_bytes_per_step': self.h2d_bytes,
             'parallelism': ('scenes' if self.kind == 'scenes' else 'attention groups') + f' x{world}'}
        if self.kind == 'nba':
            c['attention_group'] = self.G
        return c


def gather_futures_leg(head, dist, rank, world, acc, args):
    """The one collective the north star names, under the process group the bench runs on (RCCL when launched on GPUs): all-gather of the
    predicted futures [n_r, K, Tf, 2] of every rank (parallel.gather_futures; the reference's metric path wants the futures in one
    place: test.py:194,526; its only distributed code: core/utils.py:370-389).  (1) correctness: the gathered futures and the gathered
    ground truth give, on every rank, the SAME best-of-K ADE / FDE sums as the all-reduced per-rank sums of the timed run; (2) shard
    balance: agents per rank; (3) value_incl_gather: the timed region once more with the gather inside every step."""
    import torch
    from sttode_amd import parallel
    cnt = torch.zeros(world, dtype=torch.int64)
    cnt[rank] = head.n
    cnt = cnt if dist.get_backend() == 'gloo' else cnt.to(head.dev)
    dist.all_reduce(cnt)                                                             # every rank's row count: exchanged once
    counts = [int(c) for c in cnt.tolist()]
    pred_all = parallel.gather_futures(head.last_pred, counts=counts)               # [sum n_r, K, Tf, 2], rank order
    gt_all = parallel.gather_futures(head.model._future.contiguous(), counts=counts)   # [sum n_r, Tf, 2]
    ade, fde = head.model.best_of_k(pred_all, gt=gt_all)
    got = torch.stack((ade.double().sum(), fde.double().sum())).cpu()
    la, lf = head.model.best_of_k(head.last_pred, gt=head.model._future)             # the same futures, rank by rank: local sums, all-reduced
    want = torch.stack((la.double().sum(), lf.double().sum()))
    allreduce(dist, want)
    want = want.cpu()
    rel = float(((got - want).abs() / want.abs().clamp_min(1e-12)).max())
    ok = pred_all.shape[0] == sum(counts) and int(acc[2]) == sum(counts) and rel < 1e-6
    r3 = head.timed(args.steps, args.warmup, dist, 0, serial=args.serial, gather=True)   # the headline's region, with the gather inside every step
    res = {'collective': 'all_gather of the futures [n_r, K, Tf, 2] fp32, padded to the largest shard (sttode_amd.parallel.gather_futures)',
           'backend': dist.get_backend(), 'ranks': world, 'agents_per_rank': counts, 'gathered_rows': int(pred_all.shape[0]),
           'bytes_per_rank_per_step': int(head.n * K * head.Tf * 2 * 4),
           'sums_from_gathered_vs_allreduced_rel_err': rel, 'check': 'ok' if ok else 'MISMATCH',
           'gather_form': 'on a communication stream behind the call\'s event, counts exchanged once before the region, later calls launch under it',
           'ms_per_step_incl_gather': r3['ms_per_step'], 'value_incl_gather': r3['value']}
    if not ok:
        sys.stderr.write(f'bench.py: gathered futures disagree with the all-reduced metric sums on rank {rank}: {res}\n')
    return res


def train_bench(args, rank, world, dev, dist, cpu=True):
    """Training steps/s: the reference's per-scene loop (train.py:72-95: set_data with augmentation, forward, zero_grad, backward,
    Adam step) over this rank's synthetic ETH scenes; with several ranks the gradients are averaged by one flat all-reduce per
    step (sttode_amd.parallel.average_gradients).  Returns the dict (same conventions as the headline bench)."""
    import torch
    from helpers import make_args
    from sttode_amd import STTODENet, parallel, scenes
    from sttode_amd.weights import make_weights, to_torch_state_dict
    TP, TF = 8, 12
    sd = to_torch_state_dict(make_weights(1234))
    model = STTODENet(make_args('eth', TP, TF), dev)
    model.load_state_dict(sd, strict=True)
    model.train()
    # Adam as in train.py:122.  'hip' (default): sttode_amd.optim.Adam -- torch.optim.Adam's state and semantics, the step of all 88 parameters as
    # ONE HIP launch (csrc/train.hip adam_step_kernel); 'fused': torch.optim.Adam(fused=True) (three multi_tensor_apply launches of ~42 us);
    # 'foreach': torch's default form (what the reference's line constructs: ~0.3 ms of host time per step on 88 small parameters).  The
    # other two are timed on the same loop right after and reported beside it, never instead of it.
    from sttode_amd.optim import Adam as HipAdam

    def make_opt(kind):
        if kind == 'hip':
            return HipAdam(model.parameters(), lr=1e-4)
        return torch.optim.Adam(model.parameters(), lr=1e-4, **({'fused': True} if kind == 'fused' else {}))
    opt = make_opt(args.train_adam)
    nsc = args.train_scenes
    data = [scenes.eth_scene(100000 + rank * nsc + i) for i in range(nsc)]
    data = [(torch.from_numpy(o).to(dev), torch.from_numpy(p).to(dev)) for o, p in data]
    agents = sum(o.shape[0] for o, _ in data) / nsc

    TB = max(1, args.train_batch)
    if TB > 1:                                                       # batched steps: TB consecutive scenes as one CSR batch
        batches = []
        for b0 in range(0, nsc, TB):
            grp = data[b0:b0 + TB]
            ptr = torch.tensor(np.concatenate([[0], np.cumsum([o.shape[0] for o, _ in grp])]).astype('int32'), device=dev)
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

    steps = args.train_steps
    for i in range(2 * nsc):                                       # every scene size is seen twice: hipGraphs captured
        step(i)
    if dist is not None:
        dist.barrier()
    # A full pass of Python's cyclic collector over everything the earlier legs of this process left on the heap takes ~0.1 s; its
    # allocation counters used to trip inside this loop (one 100-130 ms pause in 200 steps of 0.6 ms: `ms_per_step_quarters` 0.60 / 3.18 /
    # 0.60 / 0.59, against 0.60 x 4 for `--train` alone, profiles/r05/train_leg_gc_pause.txt).  Collect now, outside the region, and move the
    # survivors out of the collector's sight: the loop then pays only for its own garbage, as it does in a process of its own.
    import gc
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for i in range(steps):
        step(i)
        if (i + 1) % max(1, steps // 4) == 0:
            marks.append(time.perf_counter())                       # (host-side marks, no synchronisation: where in the loop the time went)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    quarters = [1e3 * (b - a) / max(1, steps // 4) for a, b in zip([t0] + marks[:-1], marks)]
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        allreduce(dist, tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    out = {'metric': 'training-steps/sec (one scene per step, forward + backward + Adam)' if TB == 1 else
                     f'training-scenes/sec ({TB} scenes per step, forward + backward + Adam)', 'value': world * steps * TB / dt,
           'unit': 'steps/s' if TB == 1 else 'scenes/s', 'steps_per_s': world * steps / dt, 'n_gpus': world, 'steps': steps, 'warmup': 2 * nsc,
           'ms_per_step': 1e3 * dt / steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
           'data': 'synthetic', 'config': {'workload': f'train.py:72-95 loop over {nsc} synthetic ETH-shaped scenes per GPU (2..32 '
                                                       f'pedestrians, mean {agents:.1f}), obs={TP} pred={TF}, train() mode '
                                                       '(rotation + positional dropout), Adam lr 1e-4 (' + args.train_adam + ')',
                                           'parallelism': f'scenes x{world}' + (' + flat gradient all-reduce' if world > 1 else '')}}
    out['ms_per_step_quarters'] = [round(q, 4) for q in quarters]
    out['optimizer'] = {'hip': 'sttode_amd.optim.Adam (one HIP launch per step)', 'fused': 'torch.optim.Adam(fused=True)',
                        'foreach': 'torch.optim.Adam() (torch default, the reference\'s line)'}[args.train_adam]
    for kind in [k for k in ('fused', 'foreach') if k != args.train_adam] if args.train_adam == 'hip' else (['foreach'] if args.train_adam == 'fused' else []):
        # the same steps with torch's own optimizer forms: another optimizer on the same parameters, its own warm-up, the same timed loop
        opt = make_opt(kind)
        for i in range(nsc):
            step(i)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t0
        out[f'ms_per_step_{kind}_adam'] = 1e3 * dtf / steps
        out[f'steps_per_s_{kind}_adam'] = world * steps / dtf
    if TB == 1 and world == 1:
        out['nba_size_step'] = train_nba_step(dev, HipAdam)
    if cpu and rank == 0 and world == 1 and not args.no_cpu:
        train_cpu_baseline(args, out)
    return out


def train_nba_step(dev, Adam, steps=100):
    """The NBA branch of the same loop (train.py:59-71: set_data_nba, forward, zero_grad, backward, Adam) at the reference's batch: 32 scenes x 11
    agents per step, obs 5 / pred 10 (7 392 trajectory columns in forward()'s decoder pass)."""
    import gc
    import torch
    from helpers import make_args
    from sttode_amd import STTODENet, scenes
    from sttode_amd.weights import make_weights, to_torch_state_dict
    m = STTODENet(make_args('nba', 5, 10), dev)
    m.load_state_dict(to_torch_state_dict(make_weights(1234, past_length=5, future_length=10)), strict=True)
    m.train()
    opt = Adam(m.parameters(), lr=1e-4)
    data = []
    for i in range(4):
        d = scenes.nba_batch(900 + i, 32)
        data.append({k: (torch.from_numpy(v) if hasattr(v, 'shape') else v) for k, v in d.items()})   # the loader's host tensors

    def step(i):
        m.set_data_nba(data[i % 4])
        tot = m.forward()[0]
        opt.zero_grad()
        tot.backward()
        opt.step()
    for i in range(8):
        step(i)
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'optimizer': 'sttode_amd.optim.Adam',
            'workload': 'train.py:59-71 loop, 32 scenes x 11 agents per step, obs=5 pred=10, train() mode, host batches staged by set_data_nba'}


def train_cpu_baseline(args, out):
    """cpu_baseline of the training line: the same loop on the PyTorch-eager fp32 oracle (torch autograd) on the host cores."""
    import torch
    from helpers import make_args
    from sttode_amd import scenes
    from sttode_amd.weights import make_weights, to_torch_state_dict
    TP, TF = 8, 12
    sd = to_torch_state_dict(make_weights(1234))
    nsc = args.train_scenes
    data = [scenes.eth_scene(100000 + i) for i in range(nsc)]
    data = [(torch.from_numpy(o), torch.from_numpy(p)) for o, p in data]
    from oracle.sttode_ref import STTODENetRef                 # cpu_baseline leg only
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, ncpu)))
    ora = STTODENetRef(make_args('eth', TP, TF)).eval()
    ora.load_state_dict(sd, strict=True)
    oo = torch.optim.Adam(ora.parameters(), lr=1e-4)
    t_cpu, k = 0.0, 0
    while t_cpu < args.train_cpu_seconds or k < 2:
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
    out['speedup_vs_cpu_baseline'] = out['steps_per_s'] / out['cpu_baseline']['value']


def per_scene_leg(dev, n_scenes=256):
    """The reference's evaluation call pattern (test.py:171-188): ONE scene per call -- set_data (host tensors in, the loader's layout),
    inference (latents drawn on device), .cpu() of the futures -- in a plain loop.  Milliseconds per scene, synchronised by the D2H of every
    call; a latency figure beside the throughput headline."""
    import time
    import torch
    from helpers import make_args
    from sttode_amd import STTODENet, scenes
    from sttode_amd.weights import make_weights, to_torch_state_dict
    m = STTODENet(make_args('eth', 8, 12), dev).eval()
    m.load_state_dict(to_torch_state_dict(make_weights(1234)))
    data = [scenes.eth_scene(300000 + i) for i in range(n_scenes)]
    data = [(torch.from_numpy(o), torch.from_numpy(p)) for o, p in data]

    def loop():
        tot = 0
        for o, p in data:
            m.set_data(None, o, p, None, None)
            tot += m.inference(None).cpu().shape[1] * K
        return tot
    loop()
    best, tot = None, 0
    for _ in range(3):
        t = time.perf_counter()
        tot = loop()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    return {'ms_per_scene': 1e3 * best / n_scenes, 'trajectories_per_s': tot / best, 'scenes': n_scenes, 'agents_per_scene': '2-32',
            'pattern': 'set_data(host tensors) + inference() + .cpu() per scene, unbatched (test.py:171-188); best of 3 passes',
            'form': 'one launch of cooperating workgroups per call (csrc/scene_lat.hip)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--prewarm', type=int, default=int(os.environ.get('STTODE_BENCH_PREWARM', '40')),
                    help='extra UNTIMED steps in front of the --warmup steps of the headline region (brings the shader clock up; reported as clock_prewarm_steps)')
    ap.add_argument('--scenes', type=int, default=512, help='scenes per GPU per step of the headline workload')
    ap.add_argument('--cpu-seconds', type=float, default=8.0, help='budget of the headline CPU-baseline sample at 16 threads (half of it again at 1 thread)')
    ap.add_argument('--leg-cpu-seconds', type=float, default=1.5, help='CPU-baseline budget of each secondary leg')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--time-every', type=int, default=4, help='bracket the kernels of every n-th step with HIP events (0 = never)')
    ap.add_argument('--serial', action='store_true', help='no cross-step pipelining (one inference() per step)')
    ap.add_argument('--legs', default='all', help="secondary legs: 'all', 'none' or a comma list of " + ','.join(k for k in LEGS if k != 'eth_512'))
    ap.add_argument('--leg-steps', type=int, default=40)
    ap.add_argument('--only-leg', default='', help='run ONE leg alone (profiling passes) and print a short line')
    ap.add_argument('--train', action='store_true', help='print ONLY the training line (secondary metric: train.py:72-95 loop)')
    ap.add_argument('--no-train', action='store_true', help='skip the "train" object of the default line')
    ap.add_argument('--train-batch', type=int, default=1, help='scenes per optimizer step (1 = the reference loop)')
    ap.add_argument('--train-adam', choices=('hip', 'fused', 'foreach'), default='hip')
    ap.add_argument('--train-scenes', type=int, default=64)
    ap.add_argument('--train-steps', type=int, default=200)
    ap.add_argument('--train-cpu-seconds', type=float, default=4.0)
    ap.add_argument('--no-sustained', action='store_true', help='skip the 80-step own-stream run (key sustained)')
    ap.add_argument('--no-per-scene', action='store_true', help='skip the one-scene-per-call latency loop (key per_scene)')
    ap.add_argument('--no-exploratory', action='store_true', help='skip the exploratory bf16x3 region (key exploratory_bf16x3)')
    ap.add_argument('--no-serial-check', action='store_true', help='skip the few serial steps that give roofline.frac_serial_equivalent')
    ap.add_argument('--no-gather-futures', action='store_true', help='multi-rank runs: skip the all-gather of the futures (check + value_incl_gather)')
    ap.add_argument('--selftest-dist', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--dist-backend', choices=('nccl', 'gloo'), default='nccl', help=argparse.SUPPRESS)   # gloo: rehearsal, all ranks on cuda:0
    args = ap.parse_args()

    if args.gpus < 1:
        sys.stderr.write('bench.py: --gpus must be >= 1\n')
        return 2
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_launch(args)            # before any GPU call in this process
    if args.selftest_dist:
        return selftest_dist(args)

    import torch
    rank, world, local, dist = init_dist(args)
    if args.dist_backend == 'gloo':
        local = 0                                               # rehearsal: every rank computes on cuda:0
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    if args.train:
        out = train_bench(args, rank, world, dev, dist)
        if rank == 0:
            print(json.dumps(out))
        if dist is not None:
            dist.destroy_process_group()
        return 0

    if os.environ.get('STTODE_LAGGED') in ('2', '3', '4'):              # (experiments: the library reads the same variable)
        Leg.STREAMS = int(os.environ['STTODE_LAGGED'])
    if args.only_leg:
        # one leg alone (profiling passes: `rocprofv3 --kernel-trace --stats -- python3 bench.py --only-leg sdd_1024 --serial` gives that
        # leg's serial per-launch durations without the headline's launches of the same kernel in the table)
        if args.only_leg not in LEGS:
            sys.stderr.write(f'bench.py: unknown leg {args.only_leg!r}\n')
            return 2
        leg = Leg(args.only_leg, rank, dev, size=args.scenes if args.only_leg == 'eth_512' else None)
        lr = leg.timed(args.leg_steps, 5, dist, 1, serial=args.serial)
        lroof, lkern = leg.roofline(lr['stage_ms'], lr['value'] / world, 1, calls=None if args.serial else args.leg_steps)
        if rank == 0:
            print(json.dumps({'leg': args.only_leg, 'serial': bool(args.serial), 'value': lr['value'], 'ms_per_step': lr['ms_per_step'], 'steps': args.leg_steps,
                              'config': leg.config(world), 'roofline': lroof, 'kernels_mean_us': {k: round(v['mean_us'], 1) for k, v in lkern.items()}}))
        if dist is not None:
            dist.destroy_process_group()
        return 0
    head = Leg('eth_512', rank, dev, size=args.scenes)
    r = head.timed(args.steps, args.warmup, dist, args.time_every, serial=args.serial, prewarm=args.prewarm)
    roof, kern = head.roofline(r['stage_ms'], r['value'] / world, args.time_every, calls=None if args.serial else args.steps)
    if roof:
        roof['path_frac_survey_flops_superseded'] = r['value'] / world * F_TRAJ_SURVEY / PEAK_F32_MFMA
    if roof and not args.serial and not args.no_serial_check:
        # the same kernel with ONE launch in flight (a few serial steps, every launch bracketed): the plain per-launch formula
        # flop_per_launch / mean launch duration, the figure `rocprofv3 --kernel-trace --stats` of `bench.py --serial` reproduces
        rs = head.timed(8, 2, dist, 1, serial=True)
        # the serial form's in-launch hand-off must not have given up anywhere (its time-out word as a status; the pipelined form has none)
        head.model.native().check(head.model._workspace(head.n, head.sb.n_scenes)[0], head.n, head.sb.n_scenes)
        ms, cnt = rs['stage_ms'].get(roof['kernel'], (0.0, 0))
        if cnt:
            roof['mean_launch_s_serial'] = ms * 1e-3 / cnt
            roof['frac_serial_equivalent'] = roof['flop_per_call'] / roof['mean_launch_s_serial'] / PEAK_F32_MFMA   # (a serial launch = one call)
            roof['ms_per_step_serial'] = rs['ms_per_step']
    acc = r['metrics']
    out = {'metric': 'predicted-trajectories/sec (20-sample best-of-K)', 'value': r['value'], 'unit': 'trajectories/s',
           'n_gpus': world, 'rccl_ranks': dist.get_world_size() if dist is not None and dist.get_backend() == 'nccl' else 0,
           'dist_backend': dist.get_backend() if dist is not None else None, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': r['ms_per_step'], 'host_enqueue_ms_per_step': r['host_ms_per_step'], 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'clock_ghz': r['clock_ghz'], 'clock_prewarm_steps': args.prewarm,
           'config': head.config(world), 'roofline': roof, 'kernels': kern,
           'timed_region': 'per step: H2D of the scene batch (pinned host -> HBM), set_scene_batch, z ~ N(0,I) on device, the whole forward, '
                           'device-side best-of-K ADE/FDE; D2H of the futures excluded (value_incl_d2h includes it)',
           'step_form': f'pipelined, lagged launches over {Leg.STREAMS} streams: launch k = per-agent roles of call k (throughput form) + trajectory '
                        f'groups of call k-{Leg.STREAMS}; every step completes inside the timed region (the drain enqueues the outstanding groups)',
           'kernels_note': 'HIP-event durations on the launch streams; launches of consecutive pipelined steps share the chip, so a launch takes '
                           'longer than it would alone; a launch contains the per-agent roles of one call and the trajectory groups of another '
                           '(same shapes: its FLOP are one call\'s)'}
    if rank == 0:
        out['ade_fde_synthetic'] = [float(acc[0] / acc[2]), float(acc[1] / acc[2])]
    # D2H-inclusive figure (second key, not the headline): same steps with every step's futures copied to pinned host memory
    # (the region of the headline: as many steps, as much warm-up -- half of them made filling / draining the pipeline and the last copies
    # behind the drain weigh twice as much as they do in `value`)
    r2 = head.timed(args.steps, args.warmup, dist, 0, serial=args.serial, d2h=True)
    out['value_incl_d2h'] = r2['value']
    if not args.serial and not args.no_sustained:
        # the same workload, the same step, as a LONG run: 80 steps, so that filling and draining the pipeline and the shader clock's climb
        # after the idle gap in front of a region (~2.1 -> 2.4 GHz over 25-40 ms of load, `clock_ghz` = [start, end] of every region) weigh
        # 1/4 of what they do in the 20-step contract run.  A second figure beside `value`, never `value` itself.
        r3 = head.timed(80, 5, dist, 0)
        out['sustained'] = {'value': r3['value'], 'ms_per_step': r3['ms_per_step'], 'steps': 80, 'warmup': 5, 'clock_ghz': r3['clock_ghz'],
                            'form': 'the default step'}
    out['ms_per_step_incl_d2h'] = r2['ms_per_step']

    if not args.no_exploratory:
        # EXPLORATORY, never the headline: the same timed region with the per-trajectory chain as a three-way bf16 split on the bf16 matrix
        # cores (fp32 accumulate; STTODENet.mfma_mode = 'bf16x3').  Its own key and dtype label; parity sample below (same 1e-4 bar).
        head.model.mfma_mode = 'bf16x3'
        rx = head.timed(args.steps, args.warmup, dist, 0, serial=args.serial)
        out['exploratory_bf16x3'] = {
            'value': rx['value'], 'unit': 'trajectories/s', 'ms_per_step': rx['ms_per_step'], 'clock_ghz': rx['clock_ghz'],
            'dtype': 'bf16x3: operands of the decoder MLPs and GRU split three ways into bf16 (x = hi + mid + lo, six products per k block on '
                     'v_mfma_f32_32x32x16_bf16), fp32 accumulate, everything else f32',
            'speedup_vs_f32_headline': rx['value'] / r['value'],
            'fp32_equivalent_frac_of_fp32_mfma_peak': (rx['value'] / world) * head.F['path_per_traj'] / PEAK_F32_MFMA,
            'note': 'opt-in mode, not the product default and not `value`; held to the same golden vectors and oracle at rtol 1e-4 + atol 1e-4 '
                    '(tests/test_gpu_parity.py::test_exploratory_bf16x3_*)'}
        head.model.mfma_mode = 'f32'
    if dist is not None and not args.no_gather_futures:
        out['gather'] = gather_futures_leg(head, dist, rank, world, acc, args)
        if rank == 0:
            out['value_incl_gather'] = out['gather'].pop('value_incl_gather')

    want = [k for k in LEGS if k != 'eth_512'] if args.legs == 'all' else [] if args.legs == 'none' else args.legs.split(',')
    do_cpu = rank == 0 and world == 1 and not args.no_cpu
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    # threads: the box's usable cores, but never more than 16 -- the per-scene ops are tiny and PyTorch-CPU gets SLOWER beyond
    # that (256 threads measured 100x slower than 8); the count actually used is reported, and a 1-thread figure beside it.
    nthr = max(1, min(16, ncpu))
    # every GPU-timed region first, every CPU baseline afterwards: the oracle's OpenMP workers keep spinning after a sample and take cores
    # from the thread that enqueues the (host-paced) short legs and training steps
    legs, leg_objs = {}, {}
    for name in want:
        if name not in LEGS or name == 'eth_512':
            sys.stderr.write(f'bench.py: unknown leg {name!r}\n')
            return 2
        leg = Leg(name, rank, dev)
        # two timed regions, the faster one reported (both kept in `ms_per_step_runs`): a leg's region is only 20-80 ms long, and one
        # host pause (first use of an allocation size, a collector run of another library) moves it by tens of percent
        if not args.serial:
            leg.timed(3 * args.leg_steps, 5, dist, 0)             # (untimed: brings the shader clock up after the idle gap of building the leg)
        runs = [leg.timed(args.leg_steps, 5, dist, 1 if args.serial else 2, serial=args.serial) for _ in range(2)]
        lr = min(runs, key=lambda r: r['ms_per_step'])
        lroof, lkern = leg.roofline(lr['stage_ms'], lr['value'] / world, 2, calls=None if args.serial else args.leg_steps)
        if lroof and not args.serial and not args.no_serial_check:
            rs = leg.timed(8, 2, dist, 1, serial=True)
            ms, cnt = rs['stage_ms'].get(lroof['kernel'], (0.0, 0))
            if cnt:
                lroof['mean_launch_s_serial'] = ms * 1e-3 / cnt
                lroof['frac_serial_equivalent'] = lroof['flop_per_call'] / lroof['mean_launch_s_serial'] / PEAK_F32_MFMA
        legs[name] = {'value': lr['value'], 'unit': 'trajectories/s', 'ms_per_step': lr['ms_per_step'],
                      'host_enqueue_ms_per_step': lr['host_ms_per_step'], 'ms_per_step_runs': [r['ms_per_step'] for r in runs], 'clock_ghz': lr['clock_ghz'],
                      'steps': args.leg_steps, 'config': leg.config(world), 'roofline': lroof,
                      'kernels_mean_us': {k: round(v['mean_us'], 1) for k, v in lkern.items()}}
        leg.model.release_native()                                # packed weights / workspaces back to the allocator
        leg_objs[name] = leg
        torch.cuda.empty_cache()
    train = None
    if not args.no_train and args.legs == 'all':
        train = train_bench(args, rank, world, dev, dist, cpu=False)
    if do_cpu:
        cb, par = head.cpu_sample(args.cpu_seconds, nthr)
        cb['host_cpus_visible'] = ncpu
        cb1, _ = head.cpu_sample(args.cpu_seconds / 2, 1)
        cb['value_1_thread'] = cb1['value']
        cb['sample_1_thread'] = cb1['sample']
        out['cpu_baseline'], out['parity'] = cb, par
        out['speedup_vs_cpu_baseline'] = out['value'] / cb['value']
        if 'exploratory_bf16x3' in out:
            head.model.mfma_mode = 'bf16x3'
            _, out['exploratory_bf16x3']['parity'] = head.cpu_sample(min(2.0, args.cpu_seconds), nthr)
            head.model.mfma_mode = 'f32'
        for name, leg in leg_objs.items():
            legs[name]['cpu_baseline'], legs[name]['parity'] = leg.cpu_sample(args.leg_cpu_seconds, nthr)
    del head, leg_objs
    if legs:
        out['configs'] = legs
    if train is not None:
        if do_cpu:
            train_cpu_baseline(args, train)
        out['train'] = {k: train[k] for k in ('metric', 'steps_per_s', 'ms_per_step', 'ms_per_step_quarters', 'optimizer', 'ms_per_step_fused_adam', 'steps_per_s_fused_adam',
                                              'ms_per_step_foreach_adam', 'steps_per_s_foreach_adam', 'steps', 'config', 'nba_size_step', 'cpu_baseline',
                                              'speedup_vs_cpu_baseline') if k in train}
    if rank == 0 and world == 1 and not args.no_per_scene and not args.only_leg:
        out['per_scene'] = per_scene_leg(dev)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())

"""Upper bound of a pipelined step WITHOUT per-agent roles in the groups' launch (round 4, before building the throughput-form roles):
the per-agent tables are computed once, then every step is   H2D of the inputs -> z ~ N(0, I) -> ONE groups-only chain launch
(sttode_traj_chain, two workgroups per CU) -> best-of-K,   all on the step's own stream, `streams` streams in rotation.
    gpurun -- python profiles/exp_r04_groups_only.py [scenes] [steps] [streams] [extra_scenes]
`extra_scenes` more scenes ride along (their groups stand in for the cost of throughput-form roles: 68 workgroups of 128 agents at 512
scenes ~ 26 scenes' worth of groups); the rate is always quoted on the first `scenes` scenes' trajectories."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from sttode_amd import capi, scenes

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device('cuda:0')
base = bench.Leg('eth_512', 0, dev, size=S)
m_quote = base.m
leg = bench.Leg('eth_512', 0, dev, size=S + extra) if extra else base
K = bench.K
model = leg.model
# one serial fused call fills the workspace (tables A0x / A0y / A1y, xpad, cur, orig) of this batch
leg._load()
model.inference(None)
torch.cuda.synchronize()
buf, off = model._workspace(leg.n, leg.sb.n_scenes)
P = model.packed()['chain']
v = lambda name, cnt: buf[off[name]: off[name] + cnt]
A0x, A0y, A1y = v('A0x', leg.n * 512), v('A0y', leg.n * 512), v('A1y', leg.n * 512)
xpad, cur, orig, queue = v('xpad', leg.n * 16), v('cur', leg.n * 2), v('orig', leg.n * 2), v('queue', 64)
streams = [torch.cuda.Stream() for _ in range(NS)]
preds = [torch.empty(leg.n, K, 12, 2, device=dev) for _ in range(NS)]
inbuf = [torch.empty_like(leg.slot_bufs[0]) for _ in range(NS)]
gt = leg.model._future


def run(nsteps):
    for i in range(nsteps):
        s = streams[i % NS]
        with torch.cuda.stream(s):
            inbuf[i % NS].copy_(leg.host_buf, non_blocking=True)
            z = torch.randn(leg.n * K, 32, device=dev)
            capi.call('sttode_traj_chain', A0x, A0y, A1y, P['pool'], P['prog'], int(P['prog_len']), P['consts'], z, xpad, 16, cur, orig,
                      preds[i % NS], queue, leg.n * K, K, 8, 12, 2, s.cuda_stream)
            model.best_of_k(preds[i % NS], gt=gt)
    torch.cuda.synchronize()


run(6)
for rep in range(3):
    t = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t
    print(f'groups-only pipelined: scenes {S} (+{extra} riding along: {leg.m // 128 + 1} groups per launch), {NS} streams, {steps} steps: '
          f'{1e3 * dt / steps:.3f} ms/step = {m_quote * steps / dt / 1e6:.2f} M trajectories/s (quoted on {m_quote} trajectories)', flush=True)

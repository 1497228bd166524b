"""Who ran where and when: block-level trace of the (fused) chain launches of the pipelined headline bench (diagnostic build
lib_trace.so, C32_DIAG_TRACE: every workgroup records launch, block, kind, start / end on the 100 MHz clock, HW_ID, XCC_ID).
    VARIANTS=trace bash profiles/exp_chain_variants.sh build
    gpurun -- 'STTODE_HIP_LIB=sttode_amd/lib/variants/lib_trace.so python profiles/exp_r03_trace.py [scenes] [steps] [serial]'
Prints, per launch: start / end, blocks, median duration of role and group workgroups; over the steady-state window: mean number of
resident workgroups (of 512 slots), per-CU residency histogram, and the share of slot-time by kind."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from sttode_amd import capi
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
serial = len(sys.argv) > 3 and sys.argv[3] == 'serial'
dev = torch.device('cuda:0')
LEG = os.environ.get('LEG', 'eth_512')          # any bench leg (sdd_1024, nba_128, ...: S is then ignored unless LEG is eth_512)
leg = bench.Leg(LEG, 0, dev, size=S if LEG == 'eth_512' else None)
L = capi.lib()
L.sttode_chain_debug_buffer.argtypes = [ctypes.c_void_p]
NO_RANDN, NO_BOK = bool(os.environ.get('NO_RANDN')), bool(os.environ.get('NO_BOK'))   # diagnostic: reuse one z / skip best-of-K
if NO_RANDN or NO_BOK:
    zfix = torch.randn(leg.n * bench.K, 32, device=dev)
    def step(serial_):
        leg._load()
        h = leg.model.inference_async(z=zfix if NO_RANDN else None)
        h['gt'] = leg.model._future
        leg.pending.append(h)
        if len(leg.pending) >= leg.depth:
            hh = leg.pending.pop(0)
            pred = leg.model.wait(hh)
            if not NO_BOK:
                leg.model.best_of_k(pred.permute(1, 0, 2, 3), gt=hh['gt'])
    def drain():
        while leg.pending:
            leg.model.wait(leg.pending.pop(0))
    leg.step, leg.drain = step, drain
for _ in range(6):
    leg.step(serial)
leg.drain(); torch.cuda.synchronize()
nrec = (steps + 4) * max(4096, 16 * S)
dbg = torch.zeros(8 + nrec * 12, dtype=torch.int64, device=dev)
dbg[1] = nrec
L.sttode_chain_debug_buffer(dbg.data_ptr())
for _ in range(steps):
    leg.step(serial)
leg.drain(); torch.cuda.synchronize()
L.sttode_chain_debug_buffer(None)
d = dbg.cpu().numpy()
n = min(int(d[0]), nrec); r = d[8:8 + n * 12].reshape(n, 12)
out = os.path.join(ROOT, 'gpurun_out', 'r03_trace'); os.makedirs(out, exist_ok=True)
np.save(os.path.join(out, f'trace_s{S}_{"serial" if serial else "pipelined"}_{os.environ.get("TRACE_NAME", "x")}.npy'), r)
tag, blk, kind = r[:, 0], r[:, 1], r[:, 2]
t0, t1 = r[:, 3] * 0.01, r[:, 4] * 0.01     # us
hw, xcc = r[:, 5], r[:, 6] & 0xf
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)   # cu_id, sh_id, se_id, xcc
base = t0.min(); t0 -= base; t1 -= base
print(f'{n} records, {len(np.unique(tag))} launches, {len(np.unique(cu))} distinct CUs')
for g in np.unique(tag):
    m = tag == g
    ro, gr = m & (kind == 1), m & (kind == 0)
    gd = (t1 - t0)[gr] if gr.any() else np.zeros(1)
    print(f'launch {g:3d}: start {t0[m].min():9.1f} end {t1[m].max():9.1f} span {t1[m].max() - t0[m].min():8.1f} | roles {ro.sum():5d} median {np.median((t1 - t0)[ro]) if ro.any() else 0:7.1f} us '
          f'| groups / workers {gr.sum():5d} median {np.median(gd):7.1f} p10 {np.percentile(gd, 10):7.1f} p90 {np.percentile(gd, 90):7.1f}')
tags = np.unique(tag)
lo, hi = t0[tag == tags[len(tags) // 4]].min(), t0[tag == tags[-len(tags) // 4]].min()       # steady-state window
ev = np.concatenate([np.stack([t0, np.ones(n)], 1), np.stack([t1, -np.ones(n)], 1)]); ev = ev[np.argsort(ev[:, 0])]
occ = np.cumsum(ev[:, 1]); tt = ev[:, 0]
w = (tt[:-1] >= lo) & (tt[:-1] < hi)
mean_occ = (occ[:-1][w] * np.diff(tt)[w]).sum() / np.diff(tt)[w].sum()
print(f'steady window [{lo:.0f}, {hi:.0f}] us: mean resident chain-kernel workgroups {mean_occ:.1f} (512 slots); launches per ms {((t0 >= lo) & (t0 < hi) & (blk == 0)).sum() / (hi - lo) * 1e3:.3f}')
for k, nm in ((1, 'roles'), (0, 'groups')):
    m = (kind == k) & (t0 >= lo) & (t1 < hi)
    print(f'  {nm}: {m.sum()} blocks, slot-time {((t1 - t0)[m]).sum() / (hi - lo):.1f} slots on average, median duration {np.median((t1 - t0)[m]):.1f} us')
ro = kind == 1
if ro.any() and os.environ.get('ROLE_PHASES'):
    ph = (r[ro][:, 8:11] * 0.01 - base)
    seg = np.stack([ph[:, 0] - t0[ro], ph[:, 1] - ph[:, 0], ph[:, 2] - ph[:, 1], t1[ro] - ph[:, 2]], 1)
    print('  role phases (median us): embed %.1f | post-attention %.1f | block-0 GRU %.1f | pre-activation tables + publish %.1f' % tuple(np.median(seg, 0)))
gr = (kind == 0) & (t0 >= lo) & (t1 < hi)
clk = r[:, 7] / np.maximum((r[:, 4] - r[:, 3]) * 10.0, 1.0)      # shader cycles per ns
print(f'  shader clock over the blocks of the window: median {np.median(clk[(t0 >= lo) & (t1 < hi)]):.3f} GHz (p10 {np.percentile(clk[(t0 >= lo) & (t1 < hi)], 10):.3f}, p90 {np.percentile(clk[(t0 >= lo) & (t1 < hi)], 90):.3f})')
if (r[gr][:, 11] > 0).any():
    wt = r[gr][:, 11] * 0.01 - base - t0[gr]
    print(f'  groups: wait for tile flags: mean {wt.mean():.1f} us, median {np.median(wt):.1f}, p90 {np.percentile(wt, 90):.1f}, p99 {np.percentile(wt, 99):.1f}, max {wt.max():.1f}; '
          f'share of group slot-time {wt.sum() / (t1 - t0)[gr].sum():.3f}')
# per-CU: time-weighted distribution of resident count
res = {}
for c in np.unique(cu):
    m = cu == c
    e = np.concatenate([np.stack([t0[m], np.ones(m.sum())], 1), np.stack([t1[m], -np.ones(m.sum())], 1)]); e = e[np.argsort(e[:, 0])]
    o = np.cumsum(e[:, 1]); t = e[:, 0]; ww = (t[:-1] >= lo) & (t[:-1] < hi)
    for v in (0, 1, 2, 3):
        res[v] = res.get(v, 0.0) + (np.diff(t)[ww] * (o[:-1][ww] == v)).sum()
tot = sum(res.values())
print('  per-CU residency (share of CU-time with k chain-kernel workgroups resident):', {k: round(v / tot, 3) for k, v in res.items()})

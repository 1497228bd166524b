"""Caller-side evaluation loops over the HIP path (the reference's test.py flows, without plotting):

  eval_scenes  == test.py:163-208  (ETH/UCY/SDD): per scene set_data + inference, ADE/FDE best-of-K per agent,
                  AverageMeter weighted by agent_num  ==  plain mean over all agents of the per-agent minima.
                  Here many scenes go through ONE batched call (set_scene_batch) instead of a Python loop per scene.
  eval_nba     == test.py:495-552  (NBA): per DataLoader batch, min-over-K of the mean / final displacement at the horizons
                  1..future_length (the reference prints every 0.4 s step), weighted by batch size.
"""
import contextlib

import numpy as np
import torch


@torch.no_grad()
def eval_scenes(model, dataset, traj_scale=1.0, scenes_per_call=512, z_fn=None, pipelined=True):
    """dataset: sttode_amd.datasets.TrajectoryDataset / SDD_Dataset (or anything with ``scene_batch(indices)``).
    Returns (ADE, FDE, n_agents).  ``z_fn(n_rows)`` may supply latents (tests); otherwise torch.randn like the reference.
    ``pipelined`` (default): the calls go through ``inference_async`` -- up to four in flight, best-of-K ADE / FDE computed by the calls' own
    trajectory groups (what bench.py times) -- instead of one serial ``inference()`` + ``best_of_k`` per batch."""
    tot_a = tot_f = 0.0
    tot_n = 0
    K, zd = model.args.sample_k, model.args.zdim
    pend = []

    def finish(h):
        nonlocal tot_a, tot_f
        ade, fde = model.best_of_k_async(h, scale=traj_scale)
        model.wait(h)
        tot_a += float(ade.double().sum())
        tot_f += float(fde.double().sum())
    for s0 in range(0, len(dataset), scenes_per_call):
        sb = dataset.scene_batch(range(s0, min(s0 + scenes_per_call, len(dataset))))
        model.set_scene_batch(sb.past, sb.future, sb.scene_ptr)
        rows = sb.n_agents * K
        z = z_fn(rows) if z_fn is not None else torch.randn(rows, zd, device=model.device)
        tot_n += sb.n_agents
        if pipelined:
            if len(model._async_bufs) > 12:                               # batches of ever new sizes: per-shape slot buffers are dropped in time
                while pend:
                    finish(pend.pop(0))
                model.reset_async()
            pend.append(model.inference_async(z=z, metrics_gt=model._future, metrics_scale=traj_scale))
            if len(pend) > 4:
                finish(pend.pop(0))
            continue
        pred = model.inference(None, z=z)                                  # [K, n, Tf, 2]
        ade, fde = model.best_of_k(pred.permute(1, 0, 2, 3), scale=traj_scale)
        tot_a += float(ade.double().sum())
        tot_f += float(fde.double().sum())
    while pend:
        finish(pend.pop(0))
    if pipelined:
        model.reset_async()
    return tot_a / tot_n, tot_f / tot_n, tot_n


@torch.no_grad()
def eval_nba(model, loader, traj_scale=1.0, z_fn=None, pipelined=True, groups_per_call=16):
    """loader yields seq_collate dicts (data/dataloader_nba.py:7-18).  Returns {h: (avg_h, dest_h)} for h = 1..Tf, each the
    batch-size-weighted mean over batches of mean_n min_k (test.py:530-551).
    ``pipelined`` (default): up to ``groups_per_call`` consecutive loader batches of equal shape travel as ONE call (set_data_nba with
    [G,B,N,...]: the attention stays within each batch, exactly what the reference's one call per batch computes) through ``inference_async``
    -- several calls in flight -- and the per-horizon min-over-K metric is one HIP kernel on the call's own stream
    (``horizon_metrics_async``); the host only adds up [Tf, 2] sums.  ``pipelined=False``: one serial ``inference()`` per loader batch and
    the metric as torch ops (the round-4 loop; kept as the cross-check of tests/test_gpu_parity.py)."""
    Tf, K = model.args.future_length, model.args.sample_k
    dev = model.device
    acc = np.zeros((Tf, 2))
    count = 0
    if not pipelined:
        for data in loader:
            model.set_data_nba(data)
            n = data['past_traj'].shape[0] * data['past_traj'].shape[1]
            z = z_fn(n * K) if z_fn is not None else None
            pred = model.inference(data, z=z) * traj_scale                     # [K, n, Tf, 2]
            gt = torch.as_tensor(data['future_traj'], dtype=torch.float32).to(pred.device).reshape(n, Tf, 2) * traj_scale
            d = (pred - gt[None]).norm(dim=-1)                                 # [K, n, Tf]
            cum = d.cumsum(dim=2) / torch.arange(1, Tf + 1, device=d.device)   # mean over the first h frames
            B = data['past_traj'].shape[0]
            acc[:, 0] += cum.min(dim=0)[0].mean(dim=0).double().cpu().numpy() * B
            acc[:, 1] += d.min(dim=0)[0].mean(dim=0).double().cpu().numpy() * B
            count += B
        acc /= count
        return {h + 1: (float(acc[h, 0]), float(acc[h, 1])) for h in range(Tf)}

    pend, totals = [], []

    def finish(item):
        h, hm, B, N, G = item
        model.wait(h)
        # mean over the agents of a batch, times its batch size, summed over the call's batches == sum over all agents / N
        totals.append(hm.double().sum(dim=0) / N)

    def submit(group):
        B, N = group[0]['past_traj'].shape[:2]
        G = len(group)
        past = torch.stack([torch.as_tensor(d['past_traj'], dtype=torch.float32) for d in group])
        fut = torch.stack([torch.as_tensor(d['future_traj'], dtype=torch.float32) for d in group])
        n = G * B * N
        model.packed()
        st = model.next_async_stream(n)
        with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
            model.set_data_nba({'past_traj': past.to(dev, non_blocking=True), 'future_traj': fut.to(dev, non_blocking=True)})
            # z_fn is asked once per LOADER batch (its rows: B N K), as the serial loop asks it; the call's latents are their concatenation
            z = torch.cat([torch.as_tensor(z_fn(B * N * K)).to(dev) for _ in group]) if z_fn is not None else None
            h = model.inference_async(z=z)
        hm = model.horizon_metrics_async(h, gt=model._future, scale=traj_scale)
        pend.append((h, hm, B, N, G))
        if len(pend) > 3:
            finish(pend.pop(0))

    group = []
    for data in loader:
        shape = tuple(data['past_traj'].shape)
        if group and (tuple(group[0]['past_traj'].shape) != shape or len(group) >= groups_per_call):
            submit(group)
            group = []
        group.append(data)
        count += shape[0]
    if group:
        submit(group)
    while pend:
        finish(pend.pop(0))
    model.reset_async()
    acc = torch.stack(totals).sum(dim=0).cpu().numpy() / count
    return {h + 1: (float(acc[h, 0]), float(acc[h, 1])) for h in range(Tf)}

"""Multi-GPU: one process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).

The ETH/UCY/SDD path shards embarrassingly: scenes are independent units (pedestrians interact only through their
own scene's origin and last-agent flag; the reference's attention length is 1), so ranks take
contiguous scene ranges balanced by agent count and run the whole hot path locally with replicated weights.  There
is NO data-path collective.  Only the results are exchanged:
  * gather_futures: ONE all_gather_into_tensor of the predicted futures [n_r, K, Tf, 2] (padded to the largest shard, preallocated
    receive buffer, no count exchange when the partition is deterministic; over xGMI every rank writes its shard to its 7 peers directly), or
  * reduce_metrics: a 3-scalar all-reduce of (sum ADE, sum FDE, agents) when only metrics are needed.
NBA path: the independent unit is the forward-call batch ("attention group", e.g. 128 scenes, test.py:618): whole
groups go to ranks (shard_groups); results then match the reference exactly.
"""
import numpy as np
import torch
import torch.distributed as dist

from .scenes import shard_scenes


def shard_scene_batch(sb, rank, world):
    """This rank's contiguous slice of a SceneBatch (balanced by agent count) and its (scene0, scene1) range."""
    s0, s1 = shard_scenes(sb.scene_ptr, world)[rank]
    return sb.slice_scenes(s0, s1), (s0, s1)


def shard_groups(n_groups, rank, world):
    """NBA: contiguous range of attention groups for this rank."""
    return (n_groups * rank) // world, (n_groups * (rank + 1)) // world


def _coll_device(t, group=None):
    """Device the collective runs on: RCCL ("nccl") takes device tensors; gloo (CPU rehearsals, also with the compute on a GPU)
    gets host copies."""
    return torch.device('cpu') if dist.get_backend(group) == 'gloo' else t.device


_GATHER_BUFS = {}    # (device, dtype, world, nmax, row shape) -> ([world * nmax, ...] receive buffer, [nmax, ...] send buffer)
_GATHER_INDEX = {}   # (device, counts, nmax) -> row indices of the live rows in the padded receive buffer (rank order)


def shard_counts(scene_ptr, world):
    """Agents per rank of shard_scene_batch's partition.  shard_scenes is a pure function of (scene_ptr, world), so EVERY rank derives
    every rank's row count locally: gather_futures(..., counts=shard_counts(...)) needs no count exchange."""
    ptr = np.asarray(scene_ptr, np.int64)
    return [int(ptr[s1] - ptr[s0]) for s0, s1 in shard_scenes(ptr, world)]


def gather_futures(pred_local, group=None, counts=None, reuse=False):
    """All-gather of per-rank rows [n_r, ...] (n_r differs per rank, may be 0) -> [sum n_r, ...] in rank order, on the caller's device.

    ONE collective: ``all_gather_into_tensor`` of the shard, padded to the largest one, into a preallocated [world * nmax, ...] buffer
    (cached per shape; over xGMI every rank writes its rows to its 7 peers directly -- the result is what the reference's metric path wants
    in one place, test.py:194,526; its only distributed code: core/utils.py:370-389).  ``counts`` = every rank's row count, known on every
    rank whenever the partition is deterministic (shard_counts; equal shards of a weak-scaling run): no count exchange.  Without it the
    counts travel first as one [world] int64 all-gather.  Ragged shards are compacted by one index_select with a cached index.
    The result is a FRESH tensor by default (a caller that collects the gathered futures of successive batches, like the reference's metric
    path, must not see earlier results overwritten by the next call); ``reuse=True``: equal shards come back as a zero-copy view of the
    cached receive buffer, valid until the next call of the same shape (a timed loop that consumes each result at once).
    ``counts`` are validated the same way on EVERY rank before any rank enters the collective (length, non-negativity: properties of the
    list); the one rank-dependent check -- this rank's own row count -- fails only where the caller passed the wrong shard, a programming
    error on that rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return pred_local
    cdev = _coll_device(pred_local, group)
    if counts is None:
        mine = torch.tensor([pred_local.shape[0]], dtype=torch.int64, device=cdev)
        allc = torch.empty(world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(allc, mine, group=group)
        counts = allc.tolist()
    counts = tuple(int(c) for c in counts)
    if len(counts) != world or min(counts) < 0:
        raise ValueError(f'gather_futures: counts {counts} do not describe {world} ranks')
    if counts[dist.get_rank(group)] != pred_local.shape[0]:
        raise ValueError(f'gather_futures: counts {counts} give this rank {counts[dist.get_rank(group)]} rows, it holds {pred_local.shape[0]}')
    nmax = max(max(counts), 1)
    row = tuple(pred_local.shape[1:])
    key = (cdev, pred_local.dtype, world, nmax, row)
    if key not in _GATHER_BUFS:
        if len(_GATHER_BUFS) > 16:
            _GATHER_BUFS.clear()
        _GATHER_BUFS[key] = (torch.empty((world * nmax,) + row, dtype=pred_local.dtype, device=cdev),
                             torch.zeros((nmax,) + row, dtype=pred_local.dtype, device=cdev))
    recv, send = _GATHER_BUFS[key]
    if pred_local.shape[0] == nmax and pred_local.device == cdev and pred_local.is_contiguous():
        send = pred_local                                        # the shard itself is the send buffer
    else:
        send[: pred_local.shape[0]].copy_(pred_local)            # (rows past n_r keep their zeros / stale rows: they are never read)
    dist.all_gather_into_tensor(recv, send, group=group)
    if all(c == nmax for c in counts):
        out = recv if reuse else recv.clone()
    else:
        ikey = (cdev, counts, nmax)
        if ikey not in _GATHER_INDEX:
            if len(_GATHER_INDEX) > 64:
                _GATHER_INDEX.clear()
            _GATHER_INDEX[ikey] = torch.cat([torch.arange(r * nmax, r * nmax + c, dtype=torch.int64) for r, c in enumerate(counts)]).to(cdev)
        out = recv.index_select(0, _GATHER_INDEX[ikey])
    return out if out.device == pred_local.device else out.to(pred_local.device)


def infer_sharded(model, sb, rank, world, z=None, group=None, gather=True):
    """Scenes of ``sb`` sharded over the ranks of ``group`` (contiguous ranges balanced by agent count), the hot path run locally,
    results combined: returns (futures [K, n_total, Tf, 2] in scene order if ``gather`` else this rank's [K, n_r, Tf, 2],
    (ADE, FDE, agents) over all ranks).  ``z`` [n_total*K, zdim] (optional) is the latent of the WHOLE batch: every rank takes its rows.
    A rank whose range is empty (fewer scenes than ranks) runs no kernel and contributes zero rows / zero sums, so the
    collectives below are entered by every rank."""
    a = model.args
    K, Tf = a.sample_k, a.future_length
    local, (s0, s1) = shard_scene_batch(sb, rank, world)
    dev = model.device
    if local.n_agents > 0:
        a0 = int(sb.scene_ptr[s0])
        zl = None if z is None else torch.as_tensor(z)[a0 * K:(a0 + local.n_agents) * K]
        model.set_scene_batch(local.past, local.future, local.scene_ptr)
        pred = model.inference(None, z=zl)                                   # [K, n_r, Tf, 2]
        ade, fde = model.best_of_k(pred.permute(1, 0, 2, 3))
        sums = (ade.sum().double(), fde.sum().double())
    else:
        pred = torch.zeros(K, 0, Tf, 2, dtype=torch.float32, device=dev)
        sums = (torch.zeros((), dtype=torch.float64, device=dev), torch.zeros((), dtype=torch.float64, device=dev))
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if multi and dist.get_backend(group) == 'gloo':
        sums = tuple(t.cpu() for t in sums)
    metrics = reduce_metrics(sums[0], sums[1], local.n_agents, group=group)
    if gather and multi:
        pred = gather_futures(pred.permute(1, 0, 2, 3).contiguous(), group=group, counts=shard_counts(sb.scene_ptr, world)).permute(1, 0, 2, 3)
    return pred, metrics


def reduce_metrics(ade_sum, fde_sum, count, group=None):
    """Agent-weighted global ADE / FDE (AverageMeter(n=agent_num) semantics, test.py:205,208)."""
    t = torch.stack([torch.as_tensor(ade_sum, dtype=torch.float64).reshape(()), torch.as_tensor(fde_sum, dtype=torch.float64).reshape(()),
                     torch.as_tensor(float(count), dtype=torch.float64).reshape(())])
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if isinstance(ade_sum, torch.Tensor):
        t = t.to(_coll_device(ade_sum, group) if multi else ade_sum.device)
    if multi:
        dist.all_reduce(t, group=group)
    return float(t[0] / t[2]), float(t[1] / t[2]), int(t[2])


def average_gradients(params, group=None, weight=1.0):
    """Data-parallel training (one process per GPU, each stepping on its own scenes / NBA groups): ONE all-reduce of all gradients
    as a single flat buffer (6.5 MB for STTODENet: a single bucket over xGMI, latency- not bandwidth-bound), then the average is
    written back.  ``weight``: this rank's share (e.g. its agent count) for an agent-weighted mean; parameters whose gradient is
    None on this rank contribute zeros, so ranks may disagree on which parameters were touched (the reference has no multi-GPU
    training; this is the natural extension of its one-scene-per-step loop, train.py:72-95)."""
    params = [p for p in params if p.requires_grad]
    if not (dist.is_initialized() and dist.get_world_size(group) > 1) or not params:
        return
    dev = next((p.grad.device for p in params if p.grad is not None), params[0].device)
    flat = torch.zeros(sum(p.numel() for p in params) + 1, dtype=torch.float32, device=dev)
    off = 0
    for p in params:
        if p.grad is not None:
            flat[off: off + p.numel()] = p.grad.reshape(-1) * weight
        off += p.numel()
    flat[-1] = weight
    cdev = _coll_device(flat, group)
    if cdev != flat.device:                                       # gloo rehearsal with the compute on a GPU: the collective on a host copy
        host = flat.to(cdev)
        dist.all_reduce(host, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, group=group)
    flat[:-1] /= flat[-1]
    off = 0
    for p in params:
        g = flat[off: off + p.numel()].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()

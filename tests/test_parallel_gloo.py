"""world_size-2 gloo test (CPU): shard -> local compute -> gather equals the unsharded result, bit for bit.
Local compute is the CPU oracle here (test infrastructure); the sharding / gather / reduction logic is the product code."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import oracle_model, oracle_scene_inference


def _local_predictions(sb, z):
    ora = oracle_model('eth', 8, 12)
    outs = []
    for s in range(sb.n_scenes):
        a, b = int(sb.scene_ptr[s]), int(sb.scene_ptr[s + 1])
        obs, pred = sb.scene(s)
        outs.append(oracle_scene_inference(ora, obs, pred, z[a * 20:b * 20]).transpose(1, 0, 2, 3))  # [N,K,Tf,2]
    return np.concatenate(outs)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle.metrics_ref import best_of_k_ade_fde
    from sttode_amd import parallel, scenes
    torch.set_num_threads(2)
    sb = scenes.make_scene_batch(range(7), 'sdd')
    z = scenes.latents(5, sb.n_agents)
    local, (s0, s1) = parallel.shard_scene_batch(sb, rank, world)
    a0 = int(sb.scene_ptr[s0])
    zl = z[a0 * 20:(a0 + local.n_agents) * 20]
    pred = torch.from_numpy(_local_predictions(local, zl))
    full = parallel.gather_futures(pred, counts=parallel.shard_counts(sb.scene_ptr, world)).clone()   # ONE collective: every rank derives the counts
    assert torch.equal(parallel.gather_futures(pred), full)                      # counts exchanged instead: the same rows
    same = parallel.gather_futures(torch.full((3, 2), float(rank)), counts=[3] * world)   # equal shards: a fresh tensor by default ...
    assert same.shape == (3 * world, 2) and all(float(same[3 * r, 0]) == r for r in range(world))
    view = parallel.gather_futures(torch.full((3, 2), float(rank) + 10), counts=[3] * world, reuse=True)   # ... reuse=True: a view of the cached receive buffer
    assert all(float(view[3 * r, 0]) == r + 10 for r in range(world)) and all(float(same[3 * r, 0]) == r for r in range(world))
    again = parallel.gather_futures(torch.full((3, 2), float(rank) + 20), counts=[3] * world, reuse=True)
    assert again.data_ptr() == view.data_ptr() and float(view[0, 0]) == 20
    try:
        parallel.gather_futures(pred, counts=[1] * world)
        raise AssertionError('wrong counts were accepted')
    except ValueError:
        pass
    ade, fde = best_of_k_ade_fde(pred.numpy(), local.future)
    g = parallel.reduce_metrics(float(ade.sum()), float(fde.sum()), local.n_agents)
    if rank == 0:
        # the unsharded reference from the SAME process settings (thread count, BLAS blocking) as the sharded computation: the bitwise claim
        # is about the shard / gather logic, not about two hosts' (or two thread counts') fp32 summation orders
        q.put((full.numpy(), g, _local_predictions(sb, z)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_gather_equals_unsharded():
    from oracle.metrics_ref import best_of_k_ade_fde
    from sttode_amd import scenes
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, (ade, fde, cnt), ref = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sb = scenes.make_scene_batch(range(7), 'sdd')
    assert full.shape == ref.shape and np.array_equal(full, ref)
    # ... and the parent's own evaluation (whatever its thread count) agrees to fp32 rounding
    np.testing.assert_allclose(full, _local_predictions(sb, scenes.latents(5, sb.n_agents)), rtol=1e-4, atol=1e-4)
    ra, rf = best_of_k_ade_fde(ref, sb.future)
    assert cnt == sb.n_agents and abs(ade - ra.mean()) < 1e-6 and abs(fde - rf.mean()) < 1e-6


def test_shard_scenes_balanced_and_complete():
    from sttode_amd import scenes
    sb = scenes.make_scene_batch(range(64), 'eth')
    for world in (1, 2, 3, 8):
        parts = scenes.shard_scenes(sb.scene_ptr, world)
        assert parts[0][0] == 0 and parts[-1][1] == sb.n_scenes
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        loads = [int(sb.scene_ptr[b] - sb.scene_ptr[a]) for a, b in parts]
        assert sum(loads) == sb.n_agents and max(loads) - min(loads) <= 64


class _OracleAsModel:
    """The CPU oracle behind the few methods parallel.infer_sharded calls on a model (test infrastructure: the product model needs a GPU;
    the code under test here is the sharding / empty-rank / collective logic)."""

    def __init__(self):
        from helpers import make_args
        self.args, self.device = make_args('eth', 8, 12), torch.device('cpu')

    def set_scene_batch(self, past, future, ptr):
        from sttode_amd import scenes
        self.sb = scenes.SceneBatch(np.asarray(past), np.asarray(future), np.asarray(ptr))

    def inference(self, data=None, z=None):
        return torch.from_numpy(_local_predictions(self.sb, np.asarray(z))).permute(1, 0, 2, 3)   # [K, n, Tf, 2]

    def best_of_k(self, pred_nk):
        from oracle.metrics_ref import best_of_k_ade_fde
        a, f = best_of_k_ade_fde(pred_nk.numpy(), self.sb.future)
        return torch.from_numpy(np.asarray(a)), torch.from_numpy(np.asarray(f))


def _empty_rank_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sttode_amd import parallel, scenes
    torch.set_num_threads(2)
    sb = scenes.make_scene_batch(range(40, 42), 'sdd')            # 2 scenes, 3 ranks: rank 2 is empty
    z = scenes.latents(6, sb.n_agents)
    pred, metrics = parallel.infer_sharded(_OracleAsModel(), sb, rank, world, z=z)
    if rank == 0:
        q.put((pred.numpy(), metrics, [parallel.shard_scene_batch(sb, r, world)[0].n_agents for r in range(world)],
               _local_predictions(sb, z)))                        # (the reference from the same process settings: see _worker)
    dist.barrier()
    dist.destroy_process_group()


def test_more_ranks_than_scenes_does_not_hang():
    """world_size 3 over 2 scenes: the empty rank must enter every collective with zero rows / zero sums (round-1 advisor finding:
    it used to raise 'empty batch' locally while its peers waited in the all-gather)."""
    from oracle.metrics_ref import best_of_k_ade_fde
    from sttode_amd import scenes
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_empty_rank_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    full, (ade, fde, cnt), loads, ref = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sb = scenes.make_scene_batch(range(40, 42), 'sdd')
    assert loads[2] == 0 and loads[0] > 0 and loads[1] > 0 and sum(loads) == sb.n_agents
    assert np.array_equal(full, ref.transpose(1, 0, 2, 3))
    ra, rf = best_of_k_ade_fde(ref, sb.future)
    assert cnt == sb.n_agents and abs(ade - ra.mean()) < 1e-6 and abs(fde - rf.mean()) < 1e-6


def test_every_rank_gets_a_scene_when_there_are_enough():
    from sttode_amd import scenes
    ptr = np.array([0, 1000] + [1000 + i for i in range(1, 8)])     # one huge scene, then seven single-agent scenes
    for world in (2, 3, 8):
        parts = scenes.shard_scenes(ptr, world)
        assert all(b > a for a, b in parts) and parts[0][0] == 0 and parts[-1][1] == 8
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
    parts = scenes.shard_scenes(ptr, 12)
    assert sum(b > a for a, b in parts) == 8 and all(b == a == 8 for a, b in parts[8:])


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sttode_amd import parallel
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3), torch.nn.Linear(3, 2))
    x = torch.arange(40, dtype=torch.float32).reshape(8, 5) / 10
    xs = x[rank * 4:(rank + 1) * 4] if rank == 0 else x[4:7]                       # ranks hold 4 and 3 rows
    (net[2](torch.tanh(net[0](xs))).pow(2).sum() / xs.shape[0]).backward()         # local mean; net[3] is untouched everywhere
    if rank == 1:
        net[2].bias.grad = None                                                    # ... and one more only on this rank
    parallel.average_gradients(net.parameters(), weight=float(xs.shape[0]))
    if rank == 0:
        q.put([None if p.grad is None else p.grad.numpy().copy() for p in net.parameters()])
    dist.barrier()
    dist.destroy_process_group()


def test_average_gradients_flat_allreduce():
    """Row-weighted mean of per-rank gradients == gradient of the row-averaged objective on the union (minus the dropped bias)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3), torch.nn.Linear(3, 2))
    x = torch.arange(40, dtype=torch.float32).reshape(8, 5) / 10
    (net[2](torch.tanh(net[0](x[:7]))).pow(2).sum() / 7).backward()
    ref = [p.grad for p in net.parameters()]
    for i, (g, r) in enumerate(zip(got, ref)):
        if r is None:
            assert g is not None and not g.any()                                    # untouched everywhere: zeros after the sync
        elif i == 3:                                                                # net[2].bias: rank 1 contributed zeros
            x0 = x[:4]
            r0 = torch.autograd.grad(net[2](torch.tanh(net[0](x0))).pow(2).sum() / 7, net[2].bias)[0]
            np.testing.assert_allclose(g, r0.numpy(), rtol=1e-5, atol=1e-6)
        else:
            np.testing.assert_allclose(g, r.numpy(), rtol=1e-5, atol=1e-6)

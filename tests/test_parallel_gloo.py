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
    full = parallel.gather_futures(pred)
    ade, fde = best_of_k_ade_fde(pred.numpy(), local.future)
    g = parallel.reduce_metrics(float(ade.sum()), float(fde.sum()), local.n_agents)
    if rank == 0:
        q.put((full.numpy(), g))
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
    full, (ade, fde, cnt) = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sb = scenes.make_scene_batch(range(7), 'sdd')
    z = scenes.latents(5, sb.n_agents)
    ref = _local_predictions(sb, z)
    assert full.shape == ref.shape and np.array_equal(full, ref)
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

"""Caller-side training loop over the HIP training step: the reference's ``train()`` (train.py:55-100) and checkpoint
format (train.py:208-213), without the argparse / dataset-path plumbing.

    model = STTODENet(args, device); optimizer = Adam(model.parameters(), lr=args.lr)
    scheduler = StepLR(optimizer, step_size=args.decay_step, gamma=args.decay_gamma)
    for epoch in range(args.num_epochs):
        train_epoch(args, epoch, model, optimizer, scheduler, loader)

``model.forward()`` runs forward + (lazily, inside ``total_loss.backward()``) backward on csrc/train.hip kernels; the
optimizer is torch's (element-wise parameter update on device tensors).
"""
import torch


def train_epoch(args, epoch, model, optimizer, scheduler, train_loader, log=print, max_iters=None):
    """One epoch.  NBA: loader yields seq_collate dicts (train.py:59-71); otherwise the per-scene 10-tuples of
    TrajectoryDataset / SDD_Dataset wrapped by DataLoader(batch_size=1) (train.py:72-95).  Returns the list of total losses."""
    model.train()
    total_iter_num = len(train_loader)
    losses = []
    for iter_num, batch in enumerate(train_loader):
        if max_iters is not None and iter_num >= max_iters:
            break
        if args.dataset == 'nba':
            model.set_data_nba(batch)
        else:
            batch = list(batch)
            batch.pop()                                            # seq_name
            batch.pop()                                            # frame_idx
            obs_traj, pred_traj_gt, _, _, _, _, obs_loss_mask, pred_loss_mask = [t[0] for t in batch]
            model.set_data(batch, obs_traj, pred_traj_gt, obs_loss_mask, pred_loss_mask)
        total_loss, loss_pred, loss_recover, loss_kl, loss_diverse = model.forward()
        optimizer.zero_grad()
        total_loss.backward()
        optimizer.step()
        losses.append(float(total_loss.detach()))
        if log is not None and iter_num % getattr(args, 'iternum_print', 100) == 0:
            log('Epochs: {:02d}/{:02d}| It: {:04d}/{:04d} | Total loss: {:03f}| Loss_pred: {:03f}| Loss_recover: {:03f}| Loss_kl: {:03f}| '
                'Loss_diverse: {:03f}'.format(epoch, getattr(args, 'num_epochs', 1), iter_num, total_iter_num, losses[-1], loss_pred,
                                              loss_recover, loss_kl, loss_diverse))
    if scheduler is not None:
        scheduler.step()
    model.step_annealer()
    return losses


def save_checkpoint(path, args, model, optimizer, scheduler, epoch):
    """train.py:208-213 layout: test.py:675-678 loads ``model_cfg`` / ``model_dict`` from it."""
    torch.save({'model_dict': model.state_dict(), 'optimizer': optimizer.state_dict(),
                'scheduler': scheduler.state_dict() if scheduler is not None else None, 'epoch': epoch + 1, 'model_cfg': args}, path)


def rotate_scene_batch(past, future, scene_ptr, theta):
    """Train-mode augmentation of model/STTODE.py:419-426 for a CSR batch of scenes (``STTODENet.set_scene_batch``): scene s is
    rotated by ``theta[s]`` about its own origin (the mean of its agents' last observed positions).  past [n,Tp,2],
    future [n,Tf,2], scene_ptr [S+1], theta [S]; data preparation with torch ops on whatever device the tensors live on."""
    ptr = torch.as_tensor(scene_ptr, dtype=torch.long, device=past.device)
    counts = ptr[1:] - ptr[:-1]
    S = counts.numel()
    sid = torch.repeat_interleave(torch.arange(S, device=past.device), counts)
    orig = torch.zeros(S, 2, dtype=past.dtype, device=past.device).index_add_(0, sid, past[:, -1]) / counts[:, None].to(past.dtype)
    th = torch.as_tensor(theta, dtype=past.dtype, device=past.device)
    c, s = torch.cos(th)[sid], torch.sin(th)[sid]                     # per agent
    o = orig[sid][:, None, :]

    def rot(x):
        d = x - o
        return torch.stack((d[..., 0] * c[:, None] - d[..., 1] * s[:, None], d[..., 0] * s[:, None] + d[..., 1] * c[:, None]), dim=-1) + o
    return rot(past), (rot(future) if future is not None else None)

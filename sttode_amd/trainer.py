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

#!/usr/bin/env python3
"""Golden fixture of the reference's TRAINING STEP: run the reference itself.

Run in the build container only (needs /root/reference, read-only):
    python tests/golden/make_golden_train.py
Same stubs / deterministic weights as make_golden.py.  What is recorded (data only):

  train_step.npz
    img12, t, noise, classes, keep       one 12-channel batch (DDP:1162 layout) + the draws of GaussianDiffusion.forward
    loss                                 p_losses value (pred_x0 / l1: the DGM configuration, dgm_sample.py:41-46)
    grad.<parameter name>                d loss / d parameter from loss.backward() (CFG:770-806 through the UNet)
    grad_norm                            what clip_grad_norm_(parameters, 1.0) returns (DDP:1852)
    traj.loss                            loss of 6 consecutive optimiser steps of DDP:1840-1858 on that batch with the
                                         draws held fixed (accumulate 2, clip 1.0, Adam(lr=1e-3, betas=(0.9, 0.99)))
    traj.param_l2                        || parameters ||_2 after each of those steps
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, load_det, npz  # noqa: E402


def main():
    cfg, _ = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    m = cfg.Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    load_det(m)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective='pred_x0', loss_type='l1')
    g8 = torch.Generator().manual_seed(161)
    B = 3
    img12 = torch.rand(B, 12, 16, 16, generator=g8)
    img12[:, 6:7] = (img12[:, 6:7] > 0.4).float()
    img12[:, -2:] = (img12[:, -2:] - 0.5) * 6
    tt = torch.tensor([17, 803, 440])
    nz = torch.randn(B, 6, 16, 16, generator=g8)
    classes = torch.zeros(B, dtype=torch.long)
    data, mk, rf, fl = img12[:, :6] * 2 - 1, img12[:, 6:7], img12[:, -5:-2] * 2 - 1, img12[:, -2:]
    arrays = {'img12': img12, 't': tt, 'noise': nz, 'classes': classes}

    def loss_fn():
        torch.manual_seed(7)                                    # the class-dropout draw inside Unet.forward (CFG:422)
        return d.p_losses(data, tt, classes=classes, rgb_flow=rf, flow=fl, mask=mk, noise=nz)

    torch.manual_seed(7)
    arrays['keep'] = torch.zeros(B).float().uniform_(0, 1) < 0.5
    loss = loss_fn()
    loss.backward()
    arrays['loss'] = loss.detach()
    for k, p in m.named_parameters():
        arrays['grad.' + k] = torch.zeros_like(p) if p.grad is None else p.grad.clone()
    arrays['grad_norm'] = torch.nn.utils.clip_grad_norm_(d.parameters(), 1.0)
    m.zero_grad()

    opt = torch.optim.Adam(d.parameters(), lr=1e-3, betas=(0.9, 0.99))
    losses, pl2 = [], []
    for _ in range(6):
        total = 0.
        for _ in range(2):
            loss = loss_fn() / 2
            total += loss.item()
            loss.backward()
        torch.nn.utils.clip_grad_norm_(d.parameters(), 1.0)
        opt.step()
        opt.zero_grad()
        losses.append(total)
        pl2.append(float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in m.parameters()))))
    arrays['traj.loss'] = torch.tensor(losses, dtype=torch.float64)
    arrays['traj.param_l2'] = torch.tensor(pl2, dtype=torch.float64)
    npz('train_step', **arrays)
    print('loss', float(arrays['loss']), 'grad_norm', float(arrays['grad_norm']), 'traj', losses)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Time the training step (BASELINE config 3: dim 64, 128x128, 16 images per GPU) on the HIP kernels.

    python tools/train_bench.py [--bs 16] [--steps 5] [--warmup 2] [--accum 1] [--torch]
    torchrun --nproc-per-node N tools/train_bench.py ...        # N ranks, gradients averaged over RCCL

Prints one JSON line: images/s through a full optimiser step (forward, backward, clip, Adam, weight re-pack), and the
split forward+backward / optimiser / re-pack measured with HIP events.  ``--torch`` additionally times the same step
written with torch.nn.functional + autograd + torch.optim.Adam on the same GPU (MIOpen / rocBLAS), as a same-box
yardstick: it is the oracle's functional UNet, so it is test infrastructure, not a product path."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bs', type=int, default=16)
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--accum', type=int, default=1)
    ap.add_argument('--torch', action='store_true')
    line = run(ap.parse_args())
    if line is not None:
        print(json.dumps(line))


def run(a):
    """-> the result dict on rank 0 (None elsewhere).  ``a``: namespace with bs, size, dim, steps, warmup, accum, torch"""
    from dmhomo_amd import cfg, train, distributed as D
    from dmhomo_amd.ddpm import SyntheticConditions
    rank, world = 0, 1
    if 'RANK' in os.environ:
        rank, world, dev = D.init_from_env()
    else:
        dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(1234)
    m = cfg.Unet(dim=a.dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1).to(dev)
    d = cfg.GaussianDiffusion(m, image_size=a.size, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev)
    if world > 1:
        D.broadcast_module_(d)
    ts = train.TrainStep(d, lr=5e-4, betas=(0.9, 0.99), accum=a.accum)
    it = SyntheticConditions(a.size, a.bs, seed=1000 + rank * 100003, device=dev)
    batches = []
    for _ in range(a.accum):
        img, cls = next(it)
        img[:, :6] = torch.rand((a.bs, 6, a.size, a.size), device=dev)
        batches.append((img, cls))

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        loss = ts.step(batches)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = ts.step(batches)
    sync()
    dt = (time.perf_counter() - t0) / a.steps
    t = torch.tensor([dt], device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t)
    # split of one step (single rank view)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    acc = None
    for img, cls in batches:
        _, gr = ts.loss_and_grads(img, cls, grad_scale=1.0 / a.accum)
        acc = gr if acc is None else {k: train.ops.add(acc[k], gr[k]) for k in acc}
    ev[1].record()
    acc = ts._allreduce_mean(acc)
    ev[2].record()
    ts.apply(acc)
    ev[3].record()
    torch.cuda.synchronize()
    line = {'metric': f'training images/sec ({a.size}x{a.size}, optimiser step incl. clip + Adam + weight re-pack)',
            'value': a.bs * a.accum * world / dt, 'unit': 'images/s', 'n_gpus': world, 'ms_per_step': dt * 1e3,
            'config': {'workload': f'DGM Unet dim={a.dim}, {a.size}x{a.size}, {a.bs} images/GPU x accum {a.accum}',
                       'arithmetic': 'fp32 tensors; fp16-piece (3 MFMAs per product block) forward, data-gradient and 3x3 weight-gradient convolutions, exact-fp32 MFMA for the 1x1 / 7x7 weight gradients and the small GEMMs'},
            'split_ms': {'forward_backward': ev[0].elapsed_time(ev[1]), 'allreduce': ev[1].elapsed_time(ev[2]),
                         'clip_adam_repack': ev[2].elapsed_time(ev[3])},
            'loss': float(loss)}
    line.update({'steps': a.steps, 'warmup': a.warmup, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                 'dtype': 'f32', 'data': 'synthetic'})
    if world > 1:
        # every rank applied the same averaged gradient with the same Adam state: the parameters must agree BIT FOR BIT —
        # compared as bit patterns (rank 0's int32 view travels to every rank; a sum of sums could cancel, and NaNs compare
        # unequal to themselves), the verdict reduced with MIN over the ranks
        flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in ts.params.values()]).contiguous().view(torch.int32)
        ref = flat.clone()
        torch.distributed.broadcast(ref, src=0)
        same = torch.tensor([int(torch.equal(ref, flat))], device=flat.device, dtype=torch.int32)
        torch.distributed.all_reduce(same, op=torch.distributed.ReduceOp.MIN)
        line['ranks_agree'] = bool(int(same.item()) == 1)
        line['backend'] = torch.distributed.get_backend()
    if a.torch and rank == 0:
        line['torch_autograd_same_gpu'] = torch_leg(a, dev)
    return line if rank == 0 else None


def torch_leg(a, dev):
    """the same step in torch.nn.functional (the oracle's functional UNet) + autograd + torch.optim.Adam, fp32"""
    from oracle import unet as OU, diffusion as OD
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=a.dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    sd = {k: (v.to(dev).requires_grad_(True) if v.is_floating_point() else v.to(dev)) for k, v in m.state_dict().items()}
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.99))
    B, S = a.bs, a.size
    x = torch.randn((B, 6, S, S), device=dev)
    rf = torch.rand((B, 3, S, S), device=dev) * 2 - 1
    mk = (torch.rand((B, 1, S, S), device=dev) > 0.4).float()
    t = torch.randint(0, 1000, (B,), device=dev)
    c = torch.zeros((B,), dtype=torch.long, device=dev)
    keep = torch.rand((B,), device=dev) < 0.5

    def step():
        for _ in range(a.accum):
            out = OU.cfg_unet_forward(sd, x, t, c, rf, mk, keep)
            loss = (out - x).abs().mean() / a.accum
            loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        opt.zero_grad()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    return {'images_per_s': B * a.accum / dt, 'ms_per_step': dt * 1e3,
            'note': 'UNet forward/backward + L1 + clip + Adam only (no flow_warp term); MIOpen/rocBLAS fp32'}


if __name__ == '__main__':
    main()

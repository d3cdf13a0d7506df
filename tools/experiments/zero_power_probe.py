"""probe: the headline sampling step with all-zero weights (same kernels, same launch sequence, minimal switching power) against
random-init weights — how much of the whole step's time is the power limit.   python tools/experiments/zero_power_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmhomo_amd import cfg, ddpm, ops

dev = torch.device('cuda', 0)
for zero in (False, True, False, True):
    torch.manual_seed(0)
    model = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    model.cfg_mode = 'streams'
    d = cfg.GaussianDiffusion(model, image_size=128, timesteps=1000, sampling_timesteps=32, loss_type='l1', objective='pred_x0').to(dev)
    if zero:
        with torch.no_grad():
            for p in d.parameters():
                p.zero_()
    d.hip_graph = True
    conds = ddpm.SyntheticConditions(128, 25, seed=1000, device=dev)
    data, classes = next(conds)
    rgb_flow, flow, mask = data[:, -5:-2].contiguous(), data[:, -2:].contiguous(), data[:, -6:-5].contiguous()
    if zero:
        rgb_flow = torch.zeros_like(rgb_flow)
    d.sample(classes, rgb_flow, flow, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        d.sample(classes, rgb_flow, flow, mask)
    torch.cuda.synchronize()
    print(f'zero weights={zero}: {75 / (time.perf_counter() - t0):.2f} images/s', flush=True)

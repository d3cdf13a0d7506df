"""dev tool: is the sampling loop GPU-bound or launch-bound?  Times one bench step (a) wall with sync and (b) the
host-side time to ENQUEUE it (no sync until the end); (b) close to (a) means the CPU launch path is the limiter."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

def main():
    import argparse
    ap = argparse.ArgumentParser(); ap.add_argument('--cfg-mode', default='streams'); a = ap.parse_args()
    from dmhomo_amd import cfg
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1).to(dev)
    model.cfg_mode = a.cfg_mode
    diff = cfg.GaussianDiffusion(model, image_size=128, timesteps=1000, sampling_timesteps=32, loss_type='l1',
                                 objective='pred_x0').to(dev)
    classes = torch.zeros(25, dtype=torch.long, device=dev)
    rgb_flow = torch.rand(25, 3, 128, 128, device=dev)
    flow = torch.randn(25, 2, 128, 128, device=dev)
    mask = torch.ones(25, 1, 128, 128, device=dev)
    for _ in range(2):
        diff.sample(classes, rgb_flow, flow, mask, cond_scale=3.0); torch.cuda.synchronize()
    t0 = time.perf_counter(); diff.sample(classes, rgb_flow, flow, mask, cond_scale=3.0); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{a.cfg_mode}: enqueue {1e3 * (t1 - t0):.1f} ms, wall {1e3 * (t2 - t0):.1f} ms')
main()

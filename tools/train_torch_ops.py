"""dev tool: which torch (aten) operations — hence which small torch kernels — one optimiser step of dmhomo_amd.train issues, by
count, with the Python call sites of the most frequent ones (torch.profiler, one step after two warm-up steps):
    python tools/train_torch_ops.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity


def main():
    from dmhomo_amd import cfg, train
    from dmhomo_amd.ddpm import SyntheticConditions
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    m = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1).to(dev)
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev)
    ts = train.TrainStep(d, lr=5e-4, betas=(0.9, 0.99), accum=1)
    img, cls = next(SyntheticConditions(128, 16, seed=1000, device=dev))
    img[:, :6] = torch.rand((16, 6, 128, 128), device=dev)
    for _ in range(2):
        ts.step([(img, cls)])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        ts.step([(img, cls)])
        torch.cuda.synchronize()
    ops_, sites = collections.Counter(), collections.defaultdict(collections.Counter)
    for e in prof.events():
        if e.name.startswith('aten::') and not [c for c in e.cpu_children if c.name.startswith('aten::')]:
            ops_[e.name] += 1
            for fr in (e.stack or []):
                if 'dmhomo_amd' in fr or 'tools/' in fr:
                    sites[e.name][fr.split('dmhomo_amd/')[-1][:90]] += 1
                    break
    for name, n in ops_.most_common(14):
        print(f'{n:5d} {name}')
        for site, k in sites[name].most_common(8):
            print(f'        {k:4d}  {site}')


if __name__ == '__main__':
    main()

import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import dev, close_rel
from detweights import det_state_dict, shapes_of
from oracle import unet as OU
from dmhomo_amd import cfg, ddpm
cases = [
    dict(dim=16, dim_mults=(1, 2), channels=3, num_classes=4, resnet_block_groups=4),
    dict(dim=32, dim_mults=(1, 2, 4), channels=6, num_classes=1, init_dim=24),
    dict(dim=16, dim_mults=(1, 2, 4, 8), channels=6, num_classes=2, out_dim=5),
    dict(dim=24, dim_mults=(1, 3), channels=6, num_classes=1, resnet_block_groups=2),
    dict(dim=8, dim_mults=(1,), channels=6, num_classes=1),
    dict(dim=16, dim_mults=(1, 2), channels=6, num_classes=1, learned_variance=True),
]
for kw in cases:
    try:
        m = cfg.Unet(**kw)
        sd = det_state_dict(shapes_of(m), 5)
        m.load_state_dict(sd)
        m = m.to(dev())
        B, S = 2, 32
        g = torch.Generator().manual_seed(3)
        x = torch.randn(B, kw['channels'], S, S, generator=g)
        rf = torch.rand(B, 3, S, S, generator=g) * 2 - 1
        mk = (torch.rand(B, 1, S, S, generator=g) > 0.4).float()
        t = torch.tensor([700, 20])
        c = torch.randint(0, kw['num_classes'], (B,), generator=g)
        keep = torch.tensor([True, False])
        with torch.no_grad():
            ref = OU.cfg_unet_forward(sd, x, t, c, rf, mk, keep, groups=kw.get('resnet_block_groups', 8))
        from gpu_util import ReplayDeviceRng
        m.rng = ReplayDeviceRng([torch.where(keep, 0.25, 0.75)])
        out = m(x.to(dev()), t.to(dev()), c.to(dev()), rf.to(dev()), mk.to(dev()))
        close_rel(str(kw), out, ref, 2e-5)
        print('OK  ', kw)
    except Exception as e:
        print('FAIL', kw, type(e).__name__, str(e)[:300])

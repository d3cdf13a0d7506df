"""one-off: oracle CPU forward time vs thread count on this host (picks bench.py's cpu_baseline threads)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from oracle import unet as OU
from detweights import det_state_dict, shapes_of
from dmhomo_amd import cfg
m = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
sd = det_state_dict(shapes_of(m))
x = torch.randn(2, 6, 128, 128); rf = torch.rand(2, 3, 128, 128); mk = torch.ones(2, 1, 128, 128)
t = torch.tensor([500, 500]); c = torch.zeros(2, dtype=torch.long)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        OU.cfg_unet_forward(sd, x, t, c, rf, mk, None)
        t0 = time.perf_counter()
        for _ in range(2):
            OU.cfg_unet_forward(sd, x, t, c, rf, mk, None)
        print(nt, 'threads:', (time.perf_counter() - t0) / 2, 's per forward (bs=2)', flush=True)

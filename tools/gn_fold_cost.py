"""diagnostic (needs `make -C dmhomo_amd/csrc stamps`): what a gn_finalize folded into its CONSUMER launch would cost — the
conv kernel with an emulation of that fold at the head of every workgroup (DMH_WINO_ABLATE bit 128: 8 KB of partials -> f64
group sums -> per-channel (a, b) in LDS -> barrier), against the same build without it, on the consumer shapes of one UNet
forward; round-robin medians.  Compare the sum with what the launches cost today: 76 gn_finalize launches x 5.2-5.7 us per
denoise step, worth +1.1 % images/s when removed outright (docs/EXPERIMENTS.md, round 4).
    python tools/gn_fold_cost.py > profiles/r05_gn_fold_cost.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmhomo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops

dev = torch.device('cuda', 0)
B = 25                                             # one CFG pass of the headline (the two passes run on two streams)
# (k, C0, C1, Cout, H, launches per UNet forward): block 2 of every ResnetBlock (3x3 with the GroupNorm prologue), and the
# launch that finishes the block — the 1x1 res_conv where the width changes, else gn_silu_residual (priced as a 1x1 here)
SHAPES = [(3, 64, 0, 64, 128, 5), (3, 64, 0, 64, 64, 2), (3, 128, 0, 128, 64, 2), (3, 128, 0, 128, 32, 2),
          (3, 256, 0, 256, 32, 2), (3, 256, 0, 256, 16, 2), (3, 512, 0, 512, 16, 4),
          (1, 64, 64, 64, 128, 3), (1, 128, 64, 128, 64, 2), (1, 256, 128, 256, 32, 2), (1, 512, 256, 512, 16, 2),
          (1, 64, 0, 64, 128, 2), (1, 64, 0, 64, 64, 2), (1, 128, 0, 128, 32, 2), (1, 256, 0, 256, 16, 2), (1, 512, 0, 512, 16, 2)]


def timed(f, reps=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


tot0 = tot1 = 0.0
print(f'consumer launches of one UNet forward at B = {B} rows: without / with the emulated fold (us, median of 3 round-robin sweeps)')
for k, c0, c1, cout, H, count in SHAPES:
    w = torch.randn((cout, c0 + c1, k, k), device=dev) * (k * k * (c0 + c1)) ** -0.5
    pc = ops.PackedConv(w, torch.randn(cout, device=dev), c0, c1)
    x0 = torch.randn((B, H, H, c0), device=dev)
    x1 = torch.randn((B, H, H, c1), device=dev) if c1 else None
    coef = torch.stack([1 + 0.1 * torch.randn(B, c0, device=dev), 0.1 * torch.randn(B, c0, device=dev)], 1).contiguous()
    res = torch.randn((B, H, H, cout), device=dev) if k == 1 else None
    rcoef = torch.stack([1 + 0.1 * torch.randn(B, cout, device=dev), 0.1 * torch.randn(B, cout, device=dev)], 1).contiguous()

    def run():
        if k == 3:
            ops.conv2d(pc, x0, in_coef=coef, want_stats=True)
        else:
            ops.conv2d(pc, x0, x1, res=res, res_coef=rcoef)
    ts = {0: [], 128: []}
    for sweep in range(3):
        for abl in (0, 128):
            os.environ['DMH_WINO_ABLATE'] = str(abl)
            ts[abl].append(timed(run))
    t0, t1 = sorted(ts[0])[1], sorted(ts[128])[1]
    tot0, tot1 = tot0 + count * t0, tot1 + count * t1
    print(f'  {k}x{k} {c0 + c1:4d}->{cout:3d} @{H:3d}^2 x{count}: {t0:7.1f} -> {t1:7.1f} us  ({t1 - t0:+5.1f} per launch, {count * (t1 - t0):+6.1f} per forward)', flush=True)
os.environ['DMH_WINO_ABLATE'] = '0'
print(f'sum over one forward: {tot0:.0f} -> {tot1:.0f} us: the fold would add {tot1 - tot0:+.0f} us of kernel time per forward, '
      f'{2 * (tot1 - tot0):+.0f} us per denoise step, against 76 x 5.2..5.7 = 395..433 us of gn_finalize launches it removes')

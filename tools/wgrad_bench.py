"""dev tool: time dmh_conv_wgrad (HIP events) on training shapes; --scale multiplies dY (gradient magnitudes)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmhomo_amd import ops

SHAPES = [  # name, B, H, W, C0, C1, Cout, k, prologue
    ('3x3_64_64_128', 16, 128, 128, 64, 0, 64, 3, 0),
    ('3x3_64_64_128_pro', 16, 128, 128, 64, 0, 64, 3, 1),
    ('3x3_64+64_64_128', 16, 128, 128, 64, 64, 64, 3, 0),
    ('3x3_128_128_64_pro', 16, 64, 64, 128, 0, 128, 3, 1),
    ('3x3_256_256_32_pro', 16, 32, 32, 256, 0, 256, 3, 1),
    ('3x3_512_512_16_pro', 16, 16, 16, 512, 0, 512, 3, 1),
    ('3x3_512+512_512_16', 16, 16, 16, 512, 512, 512, 3, 0),
    ('1x1_64_384_128', 16, 128, 128, 64, 0, 384, 1, 0),
    ('1x1_128_64_128', 16, 128, 128, 128, 0, 64, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', type=str, default='')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--sparse', type=float, default=0.0, help='fraction of dY pixels set to zero')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    for name, B, H, W, C0, C1, Cout, k, pro in SHAPES:
        if a.only and a.only not in name:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        dy = torch.randn((B, H, W, Cout), device=dev, generator=g) * a.scale
        if a.sparse > 0:
            dy = dy * (torch.rand((B, H, W, 1), device=dev, generator=g) >= a.sparse)
        s0 = torch.randn((B, H, W, C0), device=dev, generator=g)
        s1 = torch.randn((B, H, W, C1), device=dev, generator=g) if C1 else None
        coef = None
        if pro:
            coef = torch.stack([1 + 0.3 * torch.randn((B, C0), device=dev, generator=g),
                                0.5 * torch.randn((B, C0), device=dev, generator=g)], 1).contiguous()
        for _ in range(3):
            ops.conv_wgrad(dy, s0, s1, k=k, in_coef=coef)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(a.reps):
            ops.conv_wgrad(dy, s0, s1, k=k, in_coef=coef)
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / a.reps * 1e3
        fl = 2.0 * B * H * W * (C0 + C1) * Cout * k * k
        print(f'{name:24s} {us:8.1f} us (wgrad + reduce)  {fl / us / 1e6:7.1f} TFLOP/s')


if __name__ == '__main__':
    main()

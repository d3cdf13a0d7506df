"""dev tool: time dmh_conv2d on the layer shapes of the BASELINE workload (HIP events), e.g.
    python tools/conv_bench.py [--reps 20] [--only 3x3_64_64_128]
also the target of the rocprofv3 --pmc passes (profiles/)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmhomo_amd import ops

SHAPES = [  # name, B, H, W, C0, C1, Cout, k, stride, ups, prologue
    ('3x3_64_64_128', 50, 128, 128, 64, 0, 64, 3, 1, 0, 1),
    ('3x3_64_64_128_nopro', 50, 128, 128, 64, 0, 64, 3, 1, 0, 0),
    ('3x3_64+64_64_128', 50, 128, 128, 64, 64, 64, 3, 1, 0, 0),
    ('3x3_128_128_64', 50, 64, 64, 128, 0, 128, 3, 1, 0, 1),
    ('3x3_256_256_32', 50, 32, 32, 256, 0, 256, 3, 1, 0, 1),
    ('3x3_512_512_16', 50, 16, 16, 512, 0, 512, 3, 1, 0, 1),
    ('3x3_256_256_16', 50, 16, 16, 256, 0, 256, 3, 1, 0, 1),
    ('3x3_64_64_64', 50, 64, 64, 64, 0, 64, 3, 1, 0, 1),
    ('3x3_128_128_32', 50, 32, 32, 128, 0, 128, 3, 1, 0, 1),
    ('3x3up_128_64_64to128', 50, 64, 64, 128, 0, 64, 3, 1, 1, 0),
    ('1x1_64_384_128', 50, 128, 128, 64, 0, 384, 1, 1, 0, 0),
    ('1x1_128_64_128', 50, 128, 128, 128, 0, 64, 1, 1, 0, 0),
    ('1x1_64+64_64_128', 50, 128, 128, 64, 64, 64, 1, 1, 0, 0),
    ('1x1_64+64_64_128_gnres', 50, 128, 128, 64, 64, 64, 1, 1, 0, 2),   # prologue field 2: + SiLU(a*res+b) residual epilogue
    ('1x1_128_512_16_res', 50, 16, 16, 128, 0, 512, 1, 1, 0, 3),        # 3: + plain residual (attention to_out)
    ('1x1_128_384_64', 50, 64, 64, 128, 0, 384, 1, 1, 0, 0),
    ('1x1_128+64_128_64_gnres', 50, 64, 64, 128, 64, 128, 1, 1, 0, 2),   # the res_convs of the deeper up-path blocks
    ('1x1_256+128_256_32_gnres', 50, 32, 32, 256, 128, 256, 1, 1, 0, 2),
    ('1x1_512+256_512_16_gnres', 50, 16, 16, 512, 256, 512, 1, 1, 0, 2),
    ('1x1_128_128_64_res', 50, 64, 64, 128, 0, 128, 1, 1, 0, 3),
    ('1x1_512_384_16', 50, 16, 16, 512, 0, 384, 1, 1, 0, 0),
    ('7x7_12_64_128', 50, 128, 128, 12, 0, 64, 7, 1, 0, 0),
    ('4x4s2_64_64_128', 50, 128, 128, 64, 0, 64, 4, 2, 0, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', type=str, default='')
    ap.add_argument('--batch', type=int, default=0, help='override the row count B')
    ap.add_argument('--bound', action='store_true', help='prologue shapes: pass DmhConv.in_bound (the block scale from a '
                    'producer-side bound instead of a search of the staged tiles)')
    ap.add_argument('--zeros', action='store_true', help='all-zero activations and weights: the same instruction stream at '
                    'minimal switching power; a large speed-up against random data means the kernel sits on the power limit')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    for name, B, H, W, C0, C1, Cout, k, stride, ups, pro in SHAPES:
        if args.only and args.only not in name:
            continue
        if args.batch:
            B = args.batch
        w = torch.randn((Cout, C0 + C1, k, k), device=dev) * (1.0 / ((C0 + C1) * k * k)) ** 0.5
        pc = ops.PackedConv(w, torch.randn(Cout, device=dev), C0, C1, stride, ups)
        s0 = torch.randn((B, H, W, C0), device=dev)
        s1 = torch.randn((B, H, W, C1), device=dev) if C1 else None
        if args.zeros:
            pc = ops.PackedConv(torch.zeros_like(w), torch.zeros(Cout, device=dev), C0, C1, stride, ups)
            s0.zero_()
            if s1 is not None:
                s1.zero_()
        coef = None
        res = rcoef = None
        if pro >= 2:
            res = torch.randn((B, H, W, Cout), device=dev)
            if pro == 2:
                rcoef = torch.stack([1 + 0.1 * torch.randn(B, Cout, device=dev), 0.1 * torch.randn(B, Cout, device=dev)], 1).contiguous()
        if pro == 1:
            coef = torch.stack([1 + 0.1 * torch.randn(B, C0, device=dev), 0.1 * torch.randn(B, C0, device=dev)], 1).contiguous()
        bound = torch.full((B, 8), 32.0, device=dev) if (args.bound and coef is not None) else None   # >= |a*x+b| here
        for _ in range(3):
            out = ops.conv2d(pc, s0, s1, in_coef=coef, res=res, res_coef=rcoef, want_stats=(k == 3 and pc.upsample2 != 2),
                             in_bound=bound)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            out = ops.conv2d(pc, s0, s1, in_coef=coef, res=res, res_coef=rcoef, want_stats=(k == 3 and pc.upsample2 != 2),
                             in_bound=bound)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.reps * 1e3
        ho, wo = out[0].shape[1:3] if isinstance(out, tuple) else out.shape[1:3]
        fl = 2.0 * k * k * (C0 + C1) * Cout * ho * wo * B
        by = 4.0 * B * (H * W * (C0 + C1) + ho * wo * Cout)
        print(f'{name:24s} {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  {by / us / 1e3:8.1f} GB/s(alg)', flush=True)


if __name__ == '__main__':
    main()

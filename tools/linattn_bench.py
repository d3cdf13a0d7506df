"""dev tool: time the fused LinearAttention passes on the layer shapes of the BASELINE workload (HIP events).
    python tools/linattn_bench.py [--rows 50]           (DMH_LIB_PATH=<other build> for an A/B on the same box)
The launches ROTATE over enough distinct inputs that their total exceeds the 256 MiB Infinity Cache (round 2 re-read one
210 MB tensor, which stayed cache resident: the PMC traffic came out BELOW the algorithmic bytes and the isolated times
were optimistic); every launch therefore reads its x from HBM, as it does inside the sampling step."""
import argparse
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmhomo_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=50)
ap.add_argument('--reps', type=int, default=24)
ap.add_argument('--only', type=int, default=-1, help='index of the one shape to run (PMC passes)')
args = ap.parse_args()
dev = torch.device('cuda', 0)
SHAPES = ((64, 128, True), (64, 64, True), (128, 64, False), (128, 32, False), (256, 32, False), (256, 16, False), (512, 16, False))
for si, (C, H, fuse_out) in enumerate(SHAPES):
    if args.only >= 0 and si != args.only:
        continue
    B = args.rows
    nbytes = B * H * H * C * 4
    nrot = max(3, -(-3 * 2 ** 28 // nbytes))          # >= 768 MiB of distinct inputs (+ the outputs written in between)
    nrot = min(nrot, 64)
    xs = [torch.randn((B, H, H, C), device=dev) for _ in range(nrot)]
    g = torch.ones(C, device=dev)
    pla = ops.PackedLinAttn(torch.randn((384, C, 1, 1), device=dev) * C ** -0.5)
    plo = ops.PackedLinAttnOut(torch.randn((64, 128, 1, 1), device=dev) * 0.1, torch.zeros(64, device=dev), torch.ones(64, device=dev)) if fuse_out else None
    for i in range(3):
        ops.linear_attention_fused(xs[i % nrot], g, pla, 32 ** -0.5, out=plo)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(args.reps):
        ops.linear_attention_fused(xs[i % nrot], g, pla, 32 ** -0.5, out=plo)
    e1.record()
    torch.cuda.synchronize()
    print(f'linattn fused C={C:3d} {H:3d}x{H:<3d} B={B} out-fused={int(fuse_out)} rotating {nrot} inputs ({nrot * nbytes / 2**20:.0f} MiB): '
          f'{e0.elapsed_time(e1) * 1000 / args.reps:8.1f} us (stats + kv + merge + qo)', flush=True)
    del xs

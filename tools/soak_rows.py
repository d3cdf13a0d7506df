"""soak: launches of the fused LinearAttention block and of the canonical conv with more workgroups than the chip holds, many
times, each compared bitwise with the same rows computed alone (timing-dependent faults show up as a few differing pixels
in some launches).      python tools/soak_rows.py [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from gpu_util import rand
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
C = 64
g = (1 + 0.2 * rand((C,), 51)).to(dev)
pla = ops.PackedLinAttn(rand((384, C, 1, 1), 52, C ** -0.5).to(dev))
plo = ops.PackedLinAttnOut((rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0).to(dev), rand((C,), 54, 0.1).to(dev), (1 + 0.2 * rand((C,), 55)).to(dev))
bad = 0
for (B, H) in ((25, 128), (50, 128), (50, 64)):
    x = (rand((B, H, H, C), 50) * 1.3 + 0.2).to(dev)
    alone = ops.linear_attention_fused(x[:2].contiguous(), g, pla, 32 ** -0.5, out=plo)
    for _ in range(N):
        y = ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo)
        bad += int(not torch.equal(y[:2], alone))
    print(f'fused LinearAttention block B={B} {H}x{H}: {N} launches, mismatching so far {bad}', flush=True)
w = rand((64, 64, 3, 3), 1, (1.0 / 576) ** 0.5).to(dev)
pc = ops.PackedConv(w, rand((64,), 2, 0.1).to(dev), 64)
x0 = rand((50, 128, 128, 64), 3).to(dev)
coef = torch.stack([1 + 0.1 * rand((50, 64), 5), 0.1 * rand((50, 64), 6)], 1).contiguous().to(dev)
a, sa = ops.conv2d(pc, x0[:2].contiguous(), in_coef=coef[:2].contiguous(), want_stats=True)
for _ in range(N):
    b, sb = ops.conv2d(pc, x0, in_coef=coef, want_stats=True)
    bad += int(not (torch.equal(b[:2], a) and torch.equal(sb[:2], sa)))
print(f'conv 3x3 64->64 @128x128 B=50 with prologue: {N} launches, mismatching in total {bad}')
sys.exit(1 if bad else 0)

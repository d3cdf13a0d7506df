import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import rand
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
C = 64
for (B, H, W) in [(25, 128, 128), (25, 64, 64), (5, 16, 16)]:
    x = (rand((B, H, W, C), 50) * 1.3 + 0.2).to(dev)
    g = (1 + 0.2 * rand((C,), 51)).to(dev)
    pla = ops.PackedLinAttn((rand((384, C, 1, 1), 52, C ** -0.5)).to(dev))
    plo = ops.PackedLinAttnOut((rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0).to(dev), rand((C,), 54, 0.1).to(dev), (1 + 0.2 * rand((C,), 55)).to(dev))
    yb = ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo)
    y2 = ops.linear_attention_fused(x[:2].contiguous(), g, pla, 32 ** -0.5, out=plo)
    y2b = ops.linear_attention_fused(x[:2].contiguous(), g, pla, 32 ** -0.5, out=plo)
    ob = ops.linear_attention_fused(x, g, pla, 32 ** -0.5)
    o2 = ops.linear_attention_fused(x[:2].contiguous(), g, pla, 32 ** -0.5)
    torch.cuda.synchronize()
    print(B, H, W, 'fused rows equal', bool(torch.equal(yb[:2], y2)), 'repeat equal', bool(torch.equal(y2, y2b)), 'max diff', float((yb[:2] - y2).abs().max()), '| core rows equal', bool(torch.equal(ob[:2], o2)))

import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import rand
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
C, H = 64, 128
g = (1 + 0.2 * rand((C,), 51)).to(dev)
pla = ops.PackedLinAttn(rand((384, C, 1, 1), 52, C ** -0.5).to(dev))
plo = ops.PackedLinAttnOut((rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0).to(dev), rand((C,), 54, 0.1).to(dev), (1 + 0.2 * rand((C,), 55)).to(dev))
x = (rand((25, H, H, C), 50) * 1.3 + 0.2).to(dev)
alone = ops.linear_attention_fused(x[:1].contiguous(), g, pla, 32 ** -0.5, out=plo)
for it in range(6):
    y = ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo)
    d = (y[:1] - alone).abs().amax(-1).reshape(-1)
    bad = (d > 1e-6).nonzero().reshape(-1)
    print(it, 'bad pixels', bad.numel(), bad[:8].tolist(), 'max diff', float(d.max()))

import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import rand
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
B, C, H, W = 1, 64, 8, 8
x = (rand((B, H, W, C), 50) * 1.3 + 0.2).to(dev)
g = (1 + 0.2 * rand((C,), 51)).to(dev)
pla = ops.PackedLinAttn((rand((384, C, 1, 1), 52, C ** -0.5)).to(dev))
plo = ops.PackedLinAttnOut((rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0).to(dev), rand((C,), 54, 0.1).to(dev), (1 + 0.2 * rand((C,), 55)).to(dev))
y = ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo)
torch.cuda.synchronize()
print('nan count', int(torch.isnan(y).sum()), 'of', y.numel(), 'inf', int(torch.isinf(y).sum()))
print(y[0, 0, 0, :8].tolist())
print(y[0, 3, 5, :8].tolist())
m = torch.isnan(y[0]).any(-1).reshape(-1).int().tolist()
print('nan pixels', [i for i, v in enumerate(m) if v])

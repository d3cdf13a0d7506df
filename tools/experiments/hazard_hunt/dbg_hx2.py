import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import rand
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
C, H = 64, 128
g = (1 + 0.2 * rand((C,), 51)).to(dev)
pla = ops.PackedLinAttn(rand((384, C, 1, 1), 52, C ** -0.5).to(dev))
plo = ops.PackedLinAttnOut((rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0).to(dev), rand((C,), 54, 0.1).to(dev), (1 + 0.2 * rand((C,), 55)).to(dev))
x = (rand((50, H, H, C), 50) * 1.3 + 0.2).to(dev)
for it in range(8):
    y = ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo).reshape(50, H * H, 16, 4)
    m = (y[..., 0] == 12345.0)
    idx = m.nonzero()
    print(it, 'marked (sample, pixel, quad) entries:', idx.shape[0], idx[:4].tolist(), [y[i[0], i[1], i[2]].tolist() for i in idx[:3]])

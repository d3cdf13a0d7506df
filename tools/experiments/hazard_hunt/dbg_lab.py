import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch.nn.functional as F
from gpu_util import rand
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().to(dev)
def nchw(t): return t.permute(0, 3, 1, 2).contiguous().cpu()
for (H, W) in [(16, 16), (7, 9), (8, 8), (16, 20), (40, 56)]:
    B, C = 3, 64
    x = rand((B, C, H, W), 50) * 1.3 + 0.2
    g = 1 + 0.2 * rand((C,), 51)
    w = rand((384, C, 1, 1), 52, C ** -0.5)
    wo = rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0
    bo = rand((C,), 54, 0.1)
    go = 1 + 0.2 * rand((C,), 55)
    def ln(t, gain):
        m = t.mean(1, keepdim=True); v = t.var(1, unbiased=False, keepdim=True)
        return (t - m) / (v + 1e-5).sqrt() * gain.double()[None, :, None, None]
    xd = x.double()
    qkv = F.conv2d(ln(xd, g), w.double())
    n = H * W
    q, k, v = [t.reshape(B, 4, 32, n) for t in qkv.chunk(3, dim=1)]
    q = q.softmax(dim=-2) * 32 ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum('b h d n, b h e n -> b h d e', k, v / n)
    core = torch.einsum('b h d e, b h d n -> b h e n', ctx, q).reshape(B, 128, H, W)
    ref = xd + ln(F.conv2d(core, wo.double(), bo.double()), go)
    pla = ops.PackedLinAttn(w.to(dev))
    plo = ops.PackedLinAttnOut(wo.to(dev), bo.to(dev), go.to(dev))
    got = nchw(ops.linear_attention_fused(nhwc(x), g.to(dev), pla, 32 ** -0.5, out=plo)).double()
    err = (got - ref).abs()
    rel = (err.max() / ref.abs().max()).item()
    e = err.reshape(B, C, n)
    print(H, W, f'rel {rel:.2e}', 'per-b', [f'{t:.1e}' for t in (e.amax((1, 2)) / ref.abs().max()).tolist()], 'frac of pixels with err > 1e-6:', float((e.amax(1) / ref.abs().max() > 1e-6).double().mean()))

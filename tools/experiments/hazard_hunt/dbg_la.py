import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch.nn.functional as F
from gpu_util import rand
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().to(dev)
def nchw(t): return t.permute(0, 3, 1, 2).contiguous().cpu()
for (C, H, W) in [(64, 16, 16), (64, 16, 20), (64, 24, 40), (64, 40, 56), (128, 24, 40)]:
    B = 3
    x = rand((B, C, H, W), 40) * 1.7 + 0.3
    x[:, :C // 2] *= 4.0
    g = 1 + 0.2 * rand((C,), 41)
    w = rand((384, C, 1, 1), 42, C ** -0.5)
    xd = x.double()
    mean = xd.mean(1, keepdim=True); var = xd.var(1, unbiased=False, keepdim=True)
    xn = (xd - mean) / (var + 1e-5).sqrt() * g.double()[None, :, None, None]
    qkv = F.conv2d(xn, w.double())
    n = H * W
    q, k, v = [t.reshape(B, 4, 32, n) for t in qkv.chunk(3, dim=1)]
    q = q.softmax(dim=-2) * 32 ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum('b h d n, b h e n -> b h d e', k, v / n)
    ref = torch.einsum('b h d e, b h d n -> b h e n', ctx, q).reshape(B, 128, H, W)
    pla = ops.PackedLinAttn(w.to(dev))
    got = nchw(ops.linear_attention_fused(nhwc(x), g.to(dev), pla, 32 ** -0.5)).double()
    err = (got - ref).abs()
    rel = (err.max() / ref.abs().max()).item()
    e = err.reshape(B, 4, 32, n)
    print(C, H, W, f'rel {rel:.2e}', 'per-b', [f'{v:.1e}' for v in (e.amax((1, 2, 3)) / ref.abs().max()).tolist()],
          'per-head', [f'{v:.1e}' for v in (e.amax((0, 2, 3)) / ref.abs().max()).tolist()],
          'per-tile', [f'{v:.0e}' for v in (e.reshape(B, 4, 32, -1)[..., :(n // 64) * 64].reshape(B, 4, 32, n // 64, 64).amax((0, 1, 2, 4)) / ref.abs().max()).tolist()][:12])
    if rel > 1e-5:
        gg = got.reshape(B, 4, 32, n); rr = ref.reshape(B, 4, 32, n)
        idx = (gg - rr).abs().argmax().item()
        bb, hh, ee, nn = idx // (4 * 32 * n), (idx // (32 * n)) % 4, (idx // n) % 32, idx % n
        terms = ctx[bb, hh, :, ee] * q[bb, hh, :, nn]
        print('   worst', (bb, hh, ee, nn), 'err', (gg - rr)[bb, hh, ee, nn].item(), 'ref', rr[bb, hh, ee, nn].item())
        print('   terms', [f'{t:.2e}' for t in terms.tolist()])
        print('   q    ', [f'{t:.2e}' for t in q[bb, hh, :, nn].tolist()])
        print('   err over e for this pixel', [f'{t:.1e}' for t in (gg - rr)[bb, hh, :, nn].tolist()])

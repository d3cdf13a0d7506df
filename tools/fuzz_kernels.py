"""randomised parity sweep of dmh_conv2d (all kernel sizes, strides, upsampling, two sources, prologue, both residual
epilogues, GroupNorm partials) and of the fused LinearAttention against fp64 references — more shapes than the fixed cases
of tests/test_gpu_kernels.py, for use after a kernel change.      python tools/fuzz_kernels.py [cases] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dmhomo_amd import ops

dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
g = torch.Generator().manual_seed(rnd.randrange(1 << 30))


def rand(shape, scale=1.0):
    return torch.randn(shape, generator=g) * scale


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(dev)


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


worst = 0.0
for it in range(N):
    kind = rnd.choice(['3x3', '3x3', '3x3', '1x1', '1x1', '7x7', '4x4s2', 'up'])
    B = rnd.randint(1, 3)
    H, W = rnd.randint(1, 40), rnd.randint(1, 40)
    c0 = rnd.choice([4, 8, 12, 20, 32, 36, 64, 96, 128, 256])
    c1 = rnd.choice([0, 0, 0, 8, 32, 64]) if kind in ('3x3', '1x1') else 0
    co = rnd.choice([4, 8, 20, 64, 72, 128, 256])
    k, stride, ups = {'3x3': (3, 1, 0), '1x1': (1, 1, 0), '7x7': (7, 1, 0), '4x4s2': (4, 2, 0), 'up': (3, 1, 1)}[kind]
    if kind == '7x7':
        c0, c1 = rnd.choice([4, 12, 16, 20, 36]), 0      # <= 16: two taps per K slice; wider: one
    if kind == '4x4s2':
        H, W = 2 * rnd.randint(1, 20), 2 * rnd.randint(1, 20)
    pro = kind in ('3x3', '1x1') and c1 == 0 and rnd.random() < 0.5
    resm = rnd.choice([0, 0, 1, 2]) if kind in ('3x3', '1x1') else 0
    stats = kind == '3x3' and rnd.random() < 0.5
    x = rand((B, c0 + c1, H, W)) * rnd.choice([1e-3, 1.0, 30.0])
    w = rand((co, c0 + c1, k, k), (1.0 / ((c0 + c1) * k * k)) ** 0.5)
    b = rand((co,), 0.3)
    xin = x.double()
    coef = None
    if pro:
        a, bb = 1 + 0.3 * rand((B, c0)), 0.5 * rand((B, c0))
        coef = torch.stack([a, bb], 1).contiguous().to(dev)
        xin = F.silu(a.double()[:, :, None, None] * xin + bb.double()[:, :, None, None])
    if ups:
        xin = F.interpolate(xin, scale_factor=2, mode='nearest')
    pad = k // 2 if stride == 1 else (1 if k == 4 else 0)
    ref = F.conv2d(xin, w.double(), b.double(), stride, pad)
    res = rcoef = None
    if resm:
        r = rand(tuple(ref.shape))
        res = nhwc(r)
        if resm == 2:
            ra, rb = 1 + 0.2 * rand((B, co)), 0.3 * rand((B, co))
            rcoef = torch.stack([ra, rb], 1).contiguous().to(dev)
            ref = ref + F.silu(ra.double()[:, :, None, None] * r.double() + rb.double()[:, :, None, None])
        else:
            ref = ref + r.double()
    pc = ops.PackedConv(w.to(dev), b.to(dev), c0, c1, stride, ups)
    xs = nhwc(x)
    s0 = xs[..., :c0].contiguous()
    s1 = xs[..., c0:].contiguous() if c1 else None
    out = ops.conv2d(pc, s0, s1, in_coef=coef, res=res, res_coef=rcoef, want_stats=stats)
    st = None
    if isinstance(out, tuple):
        out, st = out
    got = nchw(out).double()
    rel = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    ok = bool(torch.isfinite(got).all()) and rel < 4e-6
    msg = ''
    if st is not None:   # partial sums over the tiles -> per (sample, channel) sum and sum of squares
        s = st.double().sum(1).cpu()
        want = torch.stack([ref.sum((2, 3)), (ref * ref).sum((2, 3))], -1)
        srel = ((s - want).abs().max() / want.abs().max()).item()
        ok = ok and srel < 2e-5
        msg = f' stats {srel:.1e}'
    worst = max(worst, rel)
    print(f'{"ok " if ok else "BAD"} {kind:6s} B={B} {H}x{W} {c0}+{c1}->{co} pro={int(pro)} res={resm} rel={rel:.2e}{msg}', flush=True)
    if not ok:
        sys.exit(1)

for it in range(N // 4):
    C = rnd.choice([32, 64, 96, 128, 256])
    B, H, W = rnd.randint(1, 3), rnd.randint(1, 48), rnd.randint(1, 48)
    x = rand((B, C, H, W)) * rnd.choice([0.1, 1.7, 20.0]) + rnd.choice([0.0, 0.3, 5.0])
    gg = 1 + 0.2 * rand((C,))
    w = rand((384, C, 1, 1), C ** -0.5)
    xd = x.double()
    xn = (xd - xd.mean(1, keepdim=True)) / (xd.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt() * gg.double()[None, :, None, None]
    qkv = F.conv2d(xn, w.double())
    n = H * W
    q, k_, v = [t.reshape(B, 4, 32, n) for t in qkv.chunk(3, dim=1)]
    q = q.softmax(dim=-2) * 32 ** -0.5
    k_ = k_.softmax(dim=-1)
    ctx = torch.einsum('b h d n, b h e n -> b h d e', k_, v / n)
    ref = torch.einsum('b h d e, b h d n -> b h e n', ctx, q).reshape(B, 128, H, W)
    pla = ops.PackedLinAttn(w.to(dev))
    got = nchw(ops.linear_attention_fused(nhwc(x), gg.to(dev), pla, 32 ** -0.5)).double()
    rel = ((got - ref).abs().max() / ref.abs().max()).item()
    ok = bool(torch.isfinite(got).all()) and rel < 2e-5    # (the gate of tests/test_gpu_kernels.py: the fp32 LayerNorm of
    print(f'{"ok " if ok else "BAD"} linattn C={C} B={B} {H}x{W} rel={rel:.2e}', flush=True)   # offset data carries a few 1e-6 by itself)
    if not ok:
        sys.exit(1)
print(f'all {N} + {N // 4} cases within tolerance; worst conv error {worst:.2e} of the output scale')

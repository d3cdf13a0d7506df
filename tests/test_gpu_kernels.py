"""-m gpu: each HIP kernel through the C ABI against the CPU oracle / plain torch fp32 on the same inputs.

Tolerances: the kernels accumulate in fp32 like the reference but in a different order
(k-ordered MFMA fmaf chains vs oneDNN blocking), so results differ by a few fp32 ulps of the
accumulated magnitude: rtol 1e-4 / atol 1e-5 relative to unit-scale data unless stated.
Integer outputs (grid-sample corner indices, uint8 export) must be bit-exact.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev, nhwc, nchw, close, rand
from detweights import det_state_dict
from oracle import unet as OU, geometry as OG

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from dmhomo_amd import ops as _ops
    _ops.lib()
    return _ops


# ---------------------------------------------------------------------------- conv variants
CONV_CASES = [
    # name, B, H, W, C0, C1, Cout, k, stride, ups
    ('3x3 64->64 16x16', 2, 16, 16, 64, 0, 64, 3, 1, 0),
    ('3x3 ragged 8->24 19x23', 2, 19, 23, 8, 0, 24, 3, 1, 0),
    ('3x3 concat 64+32->96 20x12', 2, 20, 12, 64, 32, 96, 3, 1, 0),
    ('3x3 concat small 8+16->8 9x9', 1, 9, 9, 8, 16, 8, 3, 1, 0),
    ('3x3 upsample 32->16 7x9', 2, 7, 9, 32, 0, 16, 3, 1, 1),
    ('1x1 64->384 16x16', 2, 16, 16, 64, 0, 384, 1, 1, 0),
    ('1x1 concat 40+24->72 5x33', 3, 5, 33, 40, 24, 72, 1, 1, 0),
    ('7x7 12->64 32x32', 2, 32, 32, 12, 0, 64, 7, 1, 0),
    ('7x7 ragged 4->8 21x18', 1, 21, 18, 4, 0, 8, 7, 1, 0),
    ('7x7 16->32 20x24 (two taps per K slice, all 16 slots used)', 2, 20, 24, 16, 0, 32, 7, 1, 0),
    ('7x7 20->64 18x18 (one tap per K slice)', 1, 18, 18, 20, 0, 64, 7, 1, 0),
    ('4x4s2 64->128 32x32', 2, 32, 32, 64, 0, 128, 4, 2, 0),
    ('4x4s2 ragged 8->16 22x38', 1, 22, 38, 8, 0, 16, 4, 2, 0),
    ('2x2s2 16->32 24x40', 2, 24, 40, 16, 0, 32, 2, 2, 0),
    ('4x4s2 64->64 32x48 (space-to-depth on the fp16 cores)', 2, 32, 48, 64, 0, 64, 4, 2, 0),
    ('4x4s2 ragged 64->128 21x35', 1, 21, 35, 64, 0, 128, 4, 2, 0),
    ('4x4s2 128->256 16x16', 2, 16, 16, 128, 0, 256, 4, 2, 0),
    ('3x3 tiny 4x4 256->512', 2, 4, 4, 256, 0, 512, 3, 1, 0),
    ('3x3 upsample wide 128->128 12x20', 2, 12, 20, 128, 0, 128, 3, 1, 1),
    ('3x3 upsample sub-pixel 128->64 9x21 (ragged low-res tiles)', 2, 9, 21, 128, 0, 64, 3, 1, 1),
    ('3x3 upsample sub-pixel 512->256 16x16', 2, 16, 16, 512, 0, 256, 3, 1, 1),
    ('3x3 concat wide 128+64->128 17x9', 1, 17, 9, 128, 64, 128, 3, 1, 0),
    ('3x3 odd channels 36->20 11x13', 2, 11, 13, 36, 0, 20, 3, 1, 0),
    ('3x3 512->512 2x2 (image smaller than a tile, 16 chunks)', 3, 2, 2, 512, 0, 512, 3, 1, 0),
    ('1x1 128->64 17x9', 2, 17, 9, 128, 0, 64, 1, 1, 0),
    ('1x1 tiny 8->8 3x3', 1, 3, 3, 8, 0, 8, 1, 1, 0),
]


def ref_conv(x, w, b, k, stride, ups):
    if ups:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    pad = k // 2 if stride == 1 else (1 if k == 4 else 0)
    return F.conv2d(x, w, b, stride, pad)


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d(ops, case):
    name, B, H, W, C0, C1, Cout, k, stride, ups = case
    x = rand((B, C0 + C1, H, W), 1)
    w = rand((Cout, C0 + C1, k, k), 2, (1.0 / ((C0 + C1) * k * k)) ** 0.5)
    b = rand((Cout,), 3, 0.1)
    pc = ops.PackedConv(w.to(dev()), b.to(dev()), C0, C1, stride, ups)
    s0 = nhwc(x[:, :C0])
    s1 = nhwc(x[:, C0:]) if C1 else None
    ref = ref_conv(x, w, b, k, stride, ups)
    if pc.upsample2 == 2:
        # Upsample + conv3x3 packed in its sub-pixel form (four 2x2 convs, no GroupNorm partials): checked on its own,
        # then the classic form (one 3x3 over the virtually upsampled input) goes through the common checks below
        close(name + ' [sub-pixel]', nchw(ops.conv2d(pc, s0, s1)), ref, rtol=1e-4, atol=2e-5)
        with pytest.raises(ops._lib.DmhError):
            ops.conv2d(pc, s0, s1, want_stats=True)
        pc = ops.PackedConv(w.to(dev()), b.to(dev()), C0, C1, stride, ups, subpixel=False)
    out, stats = ops.conv2d(pc, s0, s1, want_stats=True)
    close(name, nchw(out), ref, rtol=1e-4, atol=2e-5)
    # per-tile GroupNorm partials sum to the per-channel totals
    tot = stats.sum(dim=1).cpu().double()
    close(name + ' stats.sum', tot[..., 0], ref.double().sum(dim=(2, 3)), rtol=1e-4, atol=1e-3)
    close(name + ' stats.sumsq', tot[..., 1], (ref.double() ** 2).sum(dim=(2, 3)), rtol=1e-4, atol=1e-3)


def test_conv3x3_linearity_at_bench_size(ops):
    """size-independent property at the bench's largest 3x3 geometry (50 rows of 128x128x64): conv is linear in its
    input — conv(x1 + 2 x2) - b == (conv(x1) - b) + 2 (conv(x2) - b) to fp32 accumulation accuracy, whatever block
    scales the fp16-piece kernel picks for the three inputs"""
    B, H, W, C = 50, 128, 128, 64
    w = (torch.randn((C, C, 3, 3), generator=torch.Generator().manual_seed(60)) * (1.0 / (C * 9)) ** 0.5).to(dev())
    b = (torch.randn((C,), generator=torch.Generator().manual_seed(61)) * 0.1).to(dev())
    pc = ops.PackedConv(w, b, C)
    g1, g2 = torch.Generator(device=dev()).manual_seed(62), torch.Generator(device=dev()).manual_seed(63)
    x1 = torch.randn((B, H, W, C), generator=g1, device=dev())
    x2 = torch.randn((B, H, W, C), generator=g2, device=dev()) * 3.0
    y1, y2, y12 = ops.conv2d(pc, x1), ops.conv2d(pc, x2), ops.conv2d(pc, x1 + 2 * x2)
    lhs, rhs = y12 - b, (y1 - b) + 2 * (y2 - b)
    rel = ((lhs - rhs).abs().max() / rhs.abs().max()).item()
    print(f'[parity] conv3x3 linearity at 50x128x128x64: rel_to_max={rel:.3e}')
    assert rel < 3e-6, rel


RANGE_CASES = [  # name, activation scale per 32-channel group, weight scale per output-channel half
    ('unit', (1, 1, 1, 1), (1, 1)),
    ('tiny activations', (1e-6, 1e-6, 1e-6, 1e-6), (1, 1)),
    ('huge activations', (1e6, 1e6, 1e6, 1e6), (1, 1)),
    ('growing chunks', (1e-4, 1e-2, 1, 1e3), (1, 1)),       # every chunk raises the running block maximum
    ('shrinking chunks', (1e3, 1, 1e-2, 1e-4), (1, 1)),
    ('mixed weights', (1, 1e-3, 1e2, 1), (1e-5, 1e4)),
    ('beyond fp16 range', (1e20, 1e20, 1e20, 1e20), (1e-12, 1e-12)),
]


@pytest.mark.parametrize('case', RANGE_CASES, ids=[c[0] for c in RANGE_CASES])
def test_conv3x3_dynamic_range(ops, case):
    """the 3x3 kernel keeps fp32 accuracy (error relative to the output scale, against an fp64 convolution) whatever
    the magnitudes of activations and weights — the 16-bit-piece kernels rescale blockwise, nothing may overflow"""
    name, ascale, wscale = case
    B, H, W, C, Co = 2, 24, 40, 128, 128
    x = rand((B, C, H, W), 31)
    x = x * torch.tensor(ascale, dtype=torch.float32).repeat_interleave(32)[None, :, None, None]
    w = rand((Co, C, 3, 3), 32, (1.0 / (C * 9)) ** 0.5)
    w = w * torch.tensor(wscale, dtype=torch.float32).repeat_interleave(64)[:, None, None, None]
    b = rand((Co,), 33, 0.1) * float(max(ascale)) * torch.tensor(wscale).repeat_interleave(64)
    pc = ops.PackedConv(w.to(dev()), b.to(dev()), C)
    got = nchw(ops.conv2d(pc, nhwc(x))).double()
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    # per output-channel half: error relative to that half's output scale (same bar as an fp32 fma chain, K = 1152)
    for h in range(2):
        g, r = got[:, h * 64:(h + 1) * 64], ref[:, h * 64:(h + 1) * 64]
        rel = ((g - r).abs().max() / r.abs().max()).item()
        print(f'[parity] conv3x3 range {name} half {h}: rel_to_max={rel:.3e} ref_absmax={r.abs().max().item():.3e}')
        assert torch.isfinite(g).all() and rel < 2e-6, (name, h, rel)


def test_conv2d_prologue_and_residual(ops):
    """GN-apply+SiLU prologue (zero padding applies to the ACTIVATED tensor) and both residual epilogues"""
    B, H, W, C, Co = 2, 18, 21, 32, 48
    x = rand((B, C, H, W), 4)
    w = rand((Co, C, 3, 3), 5, 0.06)
    a, bb = 1 + 0.3 * rand((B, C), 6), 0.5 * rand((B, C), 7)
    coef = torch.stack([a, bb], 1).contiguous().to(dev())                       # (B,2,C)
    res = rand((B, Co, H, W), 8)
    ra, rb = 1 + 0.2 * rand((B, Co), 9), 0.3 * rand((B, Co), 10)
    rcoef = torch.stack([ra, rb], 1).contiguous().to(dev())
    pc = ops.PackedConv(w.to(dev()), None, C)
    act = F.silu(a[:, :, None, None] * x + bb[:, :, None, None])
    base = F.conv2d(act, w, None, 1, 1)
    close('prologue', nchw(ops.conv2d(pc, nhwc(x), in_coef=coef)), base, rtol=1e-4, atol=2e-5)
    close('prologue+res', nchw(ops.conv2d(pc, nhwc(x), in_coef=coef, res=nhwc(res))), base + res, rtol=1e-4,
          atol=2e-5)
    want = base + F.silu(ra[:, :, None, None] * res + rb[:, :, None, None])
    close('prologue+gn-res', nchw(ops.conv2d(pc, nhwc(x), in_coef=coef, res=nhwc(res), res_coef=rcoef)), want,
          rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('B,H,W,C0,C1,Co,n', [(2, 18, 21, 64, 64, 64, 6), (3, 128, 128, 64, 64, 64, 6), (1, 7, 5, 32, 0, 48, 3),
                                              (2, 16, 16, 64, 0, 64, 8)])
def test_conv1x1_fused_final_projection(ops, B, H, W, C0, C1, Co, n):
    """DmhConv.fin_*: the UNet's final_conv applied inside the last res_conv launch (cat input, GroupNorm + SiLU residual) is
    bit for bit final_conv_nchw of that launch's stored output — with the output stored, and without storing it"""
    x0, x1 = rand((B, H, W, C0), 40).to(dev()), (rand((B, H, W, C1), 41).to(dev()) if C1 else None)
    pc = ops.PackedConv(rand((Co, C0 + C1, 1, 1), 42, (C0 + C1) ** -0.5).to(dev()), rand((Co,), 43, 0.1).to(dev()), C0, C1)
    res = rand((B, H, W, Co), 44).to(dev())
    rcoef = torch.stack([1 + 0.2 * rand((B, Co), 45), 0.3 * rand((B, Co), 46)], 1).contiguous().to(dev())
    fw, fb = rand((n, Co), 47, Co ** -0.5).to(dev()), rand((n,), 48, 0.2).to(dev())
    out = ops.conv2d(pc, x0, x1, res=res, res_coef=rcoef)
    want = ops.final_conv_nchw(out, fw, fb)
    out2, y = ops.conv2d(pc, x0, x1, res=res, res_coef=rcoef, final=(fw, fb))
    assert torch.equal(out2, out) and torch.equal(y, want)
    none, y2 = ops.conv2d(pc, x0, x1, res=res, res_coef=rcoef, final=(fw, None), keep_out=False)
    assert none is None and torch.equal(y2, ops.final_conv_nchw(out, fw, None))
    if Co == 64:   # DmhConv.pix_stats: the LayerNorm statistics of the output pixels, bitwise dmh_pixel_stats(out)
        out3, pst = ops.conv2d(pc, x0, x1, res=res, res_coef=rcoef, pixel_stats=True)
        want_st = torch.empty((B, H * W, 2), device=dev())
        ops.call('dmh_pixel_stats', ops.ptr(out), ops.ptr(want_st), B * H * W, Co, 1e-5, None, 0)
        assert torch.equal(out3, out) and torch.equal(pst, want_st)
    with pytest.raises(Exception, match='final projection'):
        ops.conv2d(ops.PackedConv(rand((64, 64, 3, 3), 49, 0.05).to(dev()), None, 64), x0[..., :64].contiguous() if C0 >= 64
                   else rand((B, H, W, 64), 50).to(dev()), final=(rand((n, 64), 51).to(dev()), None))


@pytest.mark.parametrize('name,C0,C1,Co,H,k', [('3x3 64->64 @128', 64, 0, 64, 128, 3), ('3x3 64+64->64 @128', 64, 64, 64, 128, 3),
                                               ('3x3 512->512 @16', 512, 0, 512, 16, 3), ('1x1 64+64->64 @128', 64, 64, 64, 128, 1)])
def test_conv2d_rows_independent_under_load(ops, name, C0, C1, Co, H, k):
    """a 50-row launch (more workgroups than the chip holds at once, two per CU) returns for its first rows — outputs and
    GroupNorm partials — bitwise what a launch of those rows alone returns, launch after launch: nothing in the kernel may
    depend on timing or on which other workgroups share the CU"""
    B = 50
    w = rand((Co, C0 + C1, k, k), 1, (1.0 / ((C0 + C1) * k * k)) ** 0.5).to(dev())
    pc = ops.PackedConv(w, rand((Co,), 2, 0.1).to(dev()), C0, C1)
    x0 = rand((B, H, H, C0), 3).to(dev())
    x1 = rand((B, H, H, C1), 4).to(dev()) if C1 else None
    coef = None
    if C1 == 0 and k == 3:
        coef = torch.stack([1 + 0.1 * rand((B, C0), 5), 0.1 * rand((B, C0), 6)], 1).contiguous().to(dev())
    res = rand((B, H, H, Co), 7).to(dev())

    def run(n):
        o = ops.conv2d(pc, x0[:n].contiguous(), None if x1 is None else x1[:n].contiguous(),
                       in_coef=None if coef is None else coef[:n].contiguous(), res=res[:n].contiguous(), want_stats=(k == 3))
        return o if isinstance(o, tuple) else (o, None)
    a, sa = run(2)
    for _ in range(3):
        b, sb = run(B)
        assert torch.equal(b[:2], a)
        assert sa is None or torch.equal(sb[:2], sa)


def test_conv2d_rejects_bad_channels(ops):
    from dmhomo_amd._lib import DmhError
    pc = ops.PackedConv(rand((8, 6, 3, 3), 1).to(dev()), None, 6)
    with pytest.raises(DmhError):
        ops.conv2d(pc, torch.zeros((1, 4, 4, 6), device=dev()))
    # the 16 x 16-pixel layout of the 3x3 kernel keeps a load's source pixel relative to its tile in 16 bits: wider images are
    # refused with a message, not mis-addressed (the 8 x 16 x 128-channel layout has no such limit)
    pc64 = ops.PackedConv(rand((64, 4, 3, 3), 2).to(dev()), None, 4)
    with pytest.raises(DmhError, match='Win <= 3853'):
        ops.conv2d(pc64, torch.zeros((1, 2, 4000, 4), device=dev()))
    ok = ops.conv2d(pc64, torch.zeros((1, 2, 3853, 4), device=dev()))
    assert ok.shape == (1, 2, 3853, 64) and float(ok.abs().max()) == 0.0


def test_ws_standardize(ops):
    w = rand((24, 40, 3, 3), 11, 0.3) + 0.05
    close('ws_fold', ops.ws_standardize(w.to(dev())).cpu(), OU.ws_fold(w), rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------- GroupNorm glue
@pytest.mark.parametrize('C,H,W', [(64, 16, 16), (8, 9, 13), (512, 4, 4)])
def test_block_gn_silu(ops, C, H, W):
    """Block.forward CFG:204-213 = conv -> gn_finalize -> SiLU(a*y+b) (+0 residual)"""
    B, groups = 2, 8
    p = det_state_dict({'proj.weight': (C, C, 3, 3), 'proj.bias': (C,), 'norm.weight': (C,), 'norm.bias': (C,)}, 2)
    x = rand((B, C, H, W), 12)
    scale, shift = rand((B, C, 1, 1), 13, 0.3), rand((B, C, 1, 1), 14, 0.3)
    ref = OU.block(p, x, groups, (scale, shift))
    ref0 = OU.block(p, x, groups)
    wn = ops.ws_standardize(p['proj.weight'].to(dev()))
    pc = ops.PackedConv(wn, p['proj.bias'].to(dev()), C)
    y, st = ops.conv2d(pc, nhwc(x), want_stats=True)
    ss = torch.cat([scale.reshape(B, C), shift.reshape(B, C)], 1).contiguous().to(dev())
    g, b = p['norm.weight'].to(dev()), p['norm.bias'].to(dev())
    coef = ops.gn_finalize(st, g, b, H * W, groups, ss)
    zero = torch.zeros_like(y)
    close(f'block C={C} ss', nchw(ops.gn_silu_residual(y, coef, zero)), ref, rtol=2e-4, atol=2e-5)
    coef0 = ops.gn_finalize(st, g, b, H * W, groups)
    close(f'block C={C}', nchw(ops.gn_silu_residual(y, coef0, None)), ref0, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize('C,H,W,B', [(64, 16, 16, 3), (64, 6, 6, 1), (128, 8, 8, 2), (128, 5, 7, 3), (256, 4, 4, 2), (256, 3, 3, 1)])
def test_gn_silu_residual_pixel_stats_bitwise(ops, C, H, W, B):
    """the per-pixel LayerNorm statistics a ResnetBlock's last kernel hands to the LinearAttention behind it are bitwise the
    ones dmh_pixel_stats computes from the block's output (ragged sizes: partially filled waves at the end of the tensor)"""
    y = rand((B, H, W, C), 41).to(dev())
    res = rand((B, H, W, C), 42).to(dev())
    coef = torch.stack([1 + 0.2 * rand((B, C), 43), 0.3 * rand((B, C), 44)], 1).contiguous().to(dev())
    plain = ops.gn_silu_residual(y, coef, res)
    out, stats = ops.gn_silu_residual(y, coef, res, pixel_stats=True)
    assert torch.equal(out, plain)
    ref = torch.empty((B, H * W, 2), device=dev())
    from dmhomo_amd._lib import call, ptr
    call('dmh_pixel_stats', ptr(out), ptr(ref), B * H * W, C, 1e-5, None, 0)
    assert torch.equal(stats, ref)
    mean = out.reshape(B, H * W, C).double().mean(-1)
    close('pixel mean', stats[..., 0].cpu().double(), mean.cpu(), rtol=1e-5, atol=1e-6)


def test_gn_silu_residual_pixel_stats_rejects_other_widths(ops):
    y = torch.zeros((1, 4, 4, 32), device=dev())
    coef = torch.zeros((1, 2, 32), device=dev())
    with pytest.raises(RuntimeError):
        ops.gn_silu_residual(y, coef, None, pixel_stats=True)


@pytest.mark.parametrize('kind', ['plain', 'offset mean', 'one outlier', 'tiny', 'huge', 'constant'])
@pytest.mark.parametrize('C,H', [(64, 24), (128, 12)])
def test_conv_prologue_static_bound(ops, kind, C, H):
    """Block 2 of a ResnetBlock (CFG:204-213) with its block scale taken from the producer's GroupNorm statistics
    (dmh_gn_finalize_bound -> DmhConv.in_bound) instead of a search of the staged tiles: the bound holds for every element
    whatever the data look like, and the convolution is the same one to fp32-accumulation accuracy"""
    B, groups = 2, 8
    y = rand((B, H, H, C), 81)
    if kind == 'offset mean':
        y = y * 0.01 + 300.0                      # |mean| >> sigma: the variance estimate cancels, the bound must not
    elif kind == 'one outlier':
        y[0, 3, 5, 7] = 4.0e4
        y[1, H - 1, H - 1, C - 1] = -9.0e3
    elif kind == 'tiny':
        y = y * 1e-18
    elif kind == 'huge':
        y = y * 1e12 + 3e11
    elif kind == 'constant':
        y = torch.full_like(y, 2.5)
    y = y.to(dev())
    # per-tile GroupNorm partials of y, as a producing conv would write them: one 'tile' holding the whole sample
    st = torch.stack([y.sum((1, 2)), (y * y).sum((1, 2))], -1).reshape(B, 1, C, 2).contiguous()
    g, b = (1 + 0.3 * rand((C,), 82)).to(dev()), (0.2 * rand((C,), 83)).to(dev())
    ss = torch.cat([0.3 * rand((B, C), 84), 0.3 * rand((B, C), 85)], 1).contiguous().to(dev())
    coef, bound = ops.gn_finalize(st, g, b, H * H, groups, ss, want_bound=True)
    assert torch.equal(coef, ops.gn_finalize(st, g, b, H * H, groups, ss))
    z = torch.nn.functional.silu(coef[:, 0].reshape(B, 1, 1, C).double() * y.double() + coef[:, 1].reshape(B, 1, 1, C).double())
    zmax = z.abs().reshape(B, -1, groups, C // groups).amax((1, 3))
    assert bool((bound.double() >= zmax).all()), (bound, zmax)       # (+inf: 'look at the data', |mean| > 64 sigma)
    assert bool(torch.isinf(bound).all()) == (kind in ('offset mean', 'constant')), bound
    w = rand((C, C, 3, 3), 86, (9 * C) ** -0.5)
    pc = ops.PackedConv(w.to(dev()), rand((C,), 87, 0.1).to(dev()), C)
    dyn = ops.conv2d(pc, y, in_coef=coef)
    sta = ops.conv2d(pc, y, in_coef=coef, in_bound=bound)
    ref = F.conv2d(z.permute(0, 3, 1, 2).cpu(), w.double(), pc.bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max().clamp_min(1e-30))
    assert bool(torch.isfinite(sta).all())
    e_dyn, e_sta = float((dyn.cpu().double() - ref).abs().max()) / scale, float((sta.cpu().double() - ref).abs().max()) / scale
    assert e_sta < 3e-6 and e_sta < 2 * e_dyn + 5e-7, (kind, e_dyn, e_sta)
    # the dynamic scale (no in_bound) bounds the PROLOGUED tile, whatever the magnitude of the raw activations is ('huge':
    # 1e12 before the GroupNorm — a block scale set by a raw value would leave the normalised ones no fp16 range)
    assert e_dyn < 3e-6, (kind, e_dyn)


def test_gn_constant_input_is_beta(ops):
    """known answer: GroupNorm of a constant tensor = beta (variance 0 -> (x-mean) = 0)"""
    B, C, H, W = 1, 16, 8, 8
    y = torch.full((B, H, W, C), 3.25, device=dev())
    st = torch.empty((B, 1, C, 2), device=dev())
    st[..., 0] = 3.25 * H * W
    st[..., 1] = 3.25 * 3.25 * H * W
    g, b = (1 + rand((C,), 1)).to(dev()), rand((C,), 2).to(dev())
    coef = ops.gn_finalize(st, g, b, H * W, 8)
    out = coef[:, 0, :] * 3.25 + coef[:, 1, :]
    close('gn const', out.cpu(), b.cpu()[None], rtol=0, atol=2e-3)      # rstd = 1/sqrt(eps) amplifies rounding


@pytest.mark.parametrize('C', [8, 64, 128, 512])
def test_chan_layernorm(ops, C):
    x = rand((2, C, 7, 9), 15) * 2 + 0.7
    g = 1 + 0.2 * rand((1, C, 1, 1), 16)
    r = rand((2, C, 7, 9), 17)
    ref = OU.chan_layernorm(x, g)
    gd = g.reshape(-1).contiguous().to(dev())
    close(f'LN C={C}', nchw(ops.chan_layernorm(nhwc(x), gd)), ref, rtol=1e-5, atol=1e-5)
    close(f'LN+res C={C}', nchw(ops.chan_layernorm(nhwc(x), gd, res=nhwc(r))), ref + r, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------- attention cores
def _split_heads(qkv):
    b, _, h, w = qkv.shape
    return [t.reshape(b, 4, 32, h * w) for t in qkv.chunk(3, dim=1)]


@pytest.mark.parametrize('H,W', [(16, 16), (4, 4), (2, 2), (13, 11), (32, 48)])
def test_linear_attention_core(ops, H, W):
    qkv = rand((2, 384, H, W), 18) * 1.5
    q, k, v = _split_heads(qkv)
    n = H * W
    q = q.softmax(dim=-2) * 32 ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum('b h d n, b h e n -> b h d e', k, v / n)
    ref = torch.einsum('b h d e, b h d n -> b h e n', ctx, q).reshape(2, 128, H, W)
    close(f'linattn {H}x{W}', nchw(ops.linear_attention_core(nhwc(qkv), 32 ** -0.5)), ref, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('C,H,W', [(64, 16, 16), (64, 7, 9), (128, 24, 40), (256, 8, 8), (64, 40, 56)])
def test_linear_attention_fused(ops, C, H, W):
    """PreNorm LayerNorm + to_qkv + LinearAttention core in two kernels (q, k, v never stored) against the unfused
    math in fp64: ragged pixel counts (sub-tiles of 64 pixels, several per workgroup), all channel widths of the UNet"""
    B = 3
    x = rand((B, C, H, W), 40) * 1.7 + 0.3
    x[:, :C // 2] *= 4.0                                   # two channel chunks with different block maxima
    g = 1 + 0.2 * rand((C,), 41)
    w = rand((384, C, 1, 1), 42, C ** -0.5)
    xd = x.double()
    mean = xd.mean(1, keepdim=True)
    var = xd.var(1, unbiased=False, keepdim=True)
    xn = (xd - mean) / (var + 1e-5).sqrt() * g.double()[None, :, None, None]
    qkv = F.conv2d(xn, w.double())
    n = H * W
    q, k, v = [t.reshape(B, 4, 32, n) for t in qkv.chunk(3, dim=1)]
    q = q.softmax(dim=-2) * 32 ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum('b h d n, b h e n -> b h d e', k, v / n)
    ref = torch.einsum('b h d e, b h d n -> b h e n', ctx, q).reshape(B, 128, H, W)
    pla = ops.PackedLinAttn(w.to(dev()))
    got = nchw(ops.linear_attention_fused(nhwc(x), g.to(dev()), pla, 32 ** -0.5)).double()
    rel = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f'[parity] linattn fused C={C} {H}x{W}: rel_to_max={rel:.3e} ref_absmax={ref.abs().max().item():.3e}')
    assert torch.isfinite(got).all() and rel < 2e-5, rel


@pytest.mark.parametrize('H,W', [(16, 16), (7, 9), (40, 56)])
def test_linear_attention_block_fused(ops, H, W):
    """dim 64: the whole Residual(PreNorm(LinearAttention)) block — LN, to_qkv, attention, to_out + bias, LN, + x — in the
    two fused passes, against fp64"""
    B, C = 3, 64
    x = rand((B, C, H, W), 50) * 1.3 + 0.2
    g = 1 + 0.2 * rand((C,), 51)
    w = rand((384, C, 1, 1), 52, C ** -0.5)
    wo = rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0          # the core output is O(1e-3): make to_out's result O(1)
    bo = rand((C,), 54, 0.1)
    go = 1 + 0.2 * rand((C,), 55)

    def ln(t, gain):
        m = t.mean(1, keepdim=True)
        v = t.var(1, unbiased=False, keepdim=True)
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
    pla = ops.PackedLinAttn(w.to(dev()))
    plo = ops.PackedLinAttnOut(wo.to(dev()), bo.to(dev()), go.to(dev()))
    got = nchw(ops.linear_attention_fused(nhwc(x), g.to(dev()), pla, 32 ** -0.5, out=plo)).double()
    rel = ((got - ref).abs().max() / ref.abs().max()).item()
    print(f'[parity] linattn block fused {H}x{W}: rel_to_max={rel:.3e} ref_absmax={ref.abs().max().item():.3e}')
    assert torch.isfinite(got).all() and rel < 2e-5, rel


def test_linear_attention_block_fused_rows_independent_under_load(ops):
    """the fused block at 128x128 with 25 samples (800 workgroups: more than the chip holds at once) returns for its first
    samples bitwise what a launch of those samples alone returns, launch after launch.  (Round 2: a variant that fed MFMA
    results straight into inline asm passed every parity case and failed exactly this — a timing-dependent read of
    accumulators the matrix core had not written yet, a few pixels per launch, only with two workgroups per CU.)"""
    C, H, W = 64, 128, 128
    g = (1 + 0.2 * rand((C,), 51)).to(dev())
    pla = ops.PackedLinAttn(rand((384, C, 1, 1), 52, C ** -0.5).to(dev()))
    plo = ops.PackedLinAttnOut((rand((C, 128, 1, 1), 53, 128 ** -0.5) * 30.0).to(dev()), rand((C,), 54, 0.1).to(dev()),
                               (1 + 0.2 * rand((C,), 55)).to(dev()))
    x = (rand((25, H, W, C), 50) * 1.3 + 0.2).to(dev())
    alone = ops.linear_attention_fused(x[:2].contiguous(), g, pla, 32 ** -0.5, out=plo)
    core = ops.linear_attention_fused(x[:2].contiguous(), g, pla, 32 ** -0.5)
    for _ in range(3):
        assert torch.equal(ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo)[:2], alone)
        assert torch.equal(ops.linear_attention_fused(x, g, pla, 32 ** -0.5)[:2], core)


def test_linear_attention_uniform_k_gives_mean_v(ops):
    """known answer: constant k -> softmax_n uniform -> ctx[d][e] = mean_n(v[e]) / n for every d (v is scaled by 1/n)"""
    H = W = 12
    qkv = rand((1, 384, H, W), 19)
    qkv[:, 128:256] = 0.37
    out = nchw(ops.linear_attention_core(nhwc(qkv), 32 ** -0.5))
    q, k, v = _split_heads(qkv)
    qs = q.softmax(dim=-2) * 32 ** -0.5                       # sums to scale over d
    want = (v.mean(-1) / (H * W))[..., None] * qs.sum(2, keepdim=True)     # (b,h,e,n)
    close('linattn uniform-k', out, want.reshape(1, 128, H, W), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('H,W', [(16, 16), (2, 2), (4, 4), (7, 9), (32, 32)])
def test_attention_core(ops, H, W):
    qkv = rand((2, 384, H, W), 20) * 1.5
    q, k, v = _split_heads(qkv)
    sim = torch.einsum('b h d i, b h d j -> b h i j', q * 32 ** -0.5, k)
    out = torch.einsum('b h i j, b h d j -> b h i d', sim.softmax(dim=-1), v)
    ref = out.permute(0, 1, 3, 2).reshape(2, 128, H, W)
    close(f'attn {H}x{W}', nchw(ops.attention_core(nhwc(qkv), 32 ** -0.5)), ref, rtol=1e-4, atol=1e-5)


def test_attention_online_softmax_rescale(ops):
    """force the running-max rescale: one late key dominates every query (guide rule 26)"""
    H = W = 16
    qkv = rand((1, 384, H, W), 21)
    qkv[:, 128:256, 13, 5] *= 25.0
    qkv[:, 0:128] = qkv[:, 128:256, 13:14, 5:6].sign() * qkv[:, 0:128].abs()
    q, k, v = _split_heads(qkv)
    sim = torch.einsum('b h d i, b h d j -> b h i j', q * 32 ** -0.5, k)
    out = torch.einsum('b h i j, b h d j -> b h i d', sim.double().softmax(dim=-1), v.double())
    ref = out.permute(0, 1, 3, 2).reshape(1, 128, H, W).float()
    close('attn rescale', nchw(ops.attention_core(nhwc(qkv), 32 ** -0.5)), ref, rtol=1e-4, atol=1e-5)


# ---------------------------------------------------------------------------- embeddings
def test_embeddings(ops):
    sd = det_state_dict({'time_mlp.1.weight': (256, 64), 'time_mlp.1.bias': (256,), 'time_mlp.3.weight': (256, 256),
                         'time_mlp.3.bias': (256,), 'classes_emb.weight': (3, 64), 'null_classes_emb': (64,),
                         'classes_mlp.0.weight': (256, 64), 'classes_mlp.0.bias': (256,),
                         'classes_mlp.2.weight': (256, 256), 'classes_mlp.2.bias': (256,)}, 3)
    t = torch.tensor([999, 0, 37, 500])
    import math
    half = 32
    freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    se = ops.sinusoidal_embed(t.to(dev()), freq.to(dev()))
    close('sinusoid', se.cpu(), OU.sinusoidal_pos_emb(t, 64), rtol=0, atol=2e-6)
    d = lambda k: sd[k].to(dev())
    h = ops.linear(se, d('time_mlp.1.weight').t().contiguous(), d('time_mlp.1.bias'), act_out='gelu')
    te = ops.linear(h, d('time_mlp.3.weight').t().contiguous(), d('time_mlp.3.bias'))
    close('time_mlp', te.cpu(), OU.time_mlp(sd, t, 64), rtol=1e-4, atol=1e-5)
    classes = torch.tensor([0, 2, 1, 2])
    keep = torch.tensor([True, False, True, False])
    ce = ops.class_embed(classes.to(dev()), keep.to(torch.uint8).to(dev()), d('classes_emb.weight'),
                         d('null_classes_emb'))
    hm = ops.linear(ce, d('classes_mlp.0.weight').t().contiguous(), d('classes_mlp.0.bias'), act_out='gelu')
    cm = ops.linear(hm, d('classes_mlp.2.weight').t().contiguous(), d('classes_mlp.2.bias'))
    close('classes_mlp', cm.cpu(), OU.class_mlp(sd, classes, keep), rtol=1e-4, atol=1e-5)
    w = rand((640, 512), 22, 0.05)
    x = rand((4, 512), 23)
    y = ops.linear(x.to(dev()), w.t().contiguous().to(dev()), None, act_in='silu')
    close('mlp silu-in', y.cpu(), F.linear(F.silu(x), w), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('R,i,o', [(7, 512, 200), (50, 96, 1000), (5, 1, 70), (9, 333, 64)])
def test_linear_rows_independent_and_ragged(ops, R, i, o):
    """dmh_linear shares each weight load between the rows of a block: a row's result must not depend on the rows beside it
    (ragged row counts and feature counts that are no multiple of the block's 4 rows / 64 columns / 4 input quarters)"""
    x = rand((R, i), 71).to(dev())
    wt = rand((i, o), 72, i ** -0.5).to(dev())
    b = rand((o,), 73).to(dev())
    y = ops.linear(x, wt, b, act_in='silu', act_out='gelu')
    ref = F.gelu(F.linear(F.silu(x.double().cpu()), wt.double().cpu().t(), b.double().cpu()))
    close(f'linear {R}x{i}->{o}', y.cpu().double(), ref, rtol=1e-5, atol=2e-6)
    for r in (0, R // 2, R - 1):
        alone = ops.linear(x[r:r + 1].contiguous(), wt, b, act_in='silu', act_out='gelu')
        assert torch.equal(alone[0], y[r]), r
    tail = ops.linear(x[R - 3:].contiguous(), wt, b, act_in='silu', act_out='gelu')
    assert torch.equal(tail, y[R - 3:])


# ---------------------------------------------------------------------------- sampler glue
def test_assemble_and_final_conv(ops):
    x, rf, m = rand((2, 6, 9, 11), 24), rand((2, 3, 9, 11), 25), (rand((2, 1, 9, 11), 26) > 0).float()
    out = ops.assemble_input(x.to(dev()), rf.to(dev()), m.to(dev()), reps=2, cpad=12)
    want = torch.cat([x, rf * m, torch.zeros(2, 3, 9, 11)], 1).repeat(2, 1, 1, 1)
    assert torch.equal(nchw(out), want)
    h = rand((3, 64, 5, 7), 27)
    w, b = rand((6, 64), 28, 0.1), rand((6,), 29)
    got = ops.final_conv_nchw(nhwc(h), w.to(dev()), b.to(dev()))
    close('final_conv', got.cpu(), F.conv2d(h, w[:, :, None, None], b), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('objective', ['pred_noise', 'pred_x0', 'pred_v'])
def test_sampler_step(ops, objective):
    from dmhomo_amd._lib import DmhStep
    from oracle import diffusion as OD
    buf = OD.schedule_buffers(1000, 'cosine')
    n = (2, 6, 8, 8)
    cond, null, x, noise = (rand(n, s) for s in (30, 31, 32, 33))
    t, tn, s = 749, 499, 3.0
    mo = null + (cond - null) * s
    tt = torch.full((2,), t, dtype=torch.long)
    pn, x0 = OD._predictions(buf, objective, mo, x, tt, True)
    want = OD._ddim_update(buf, x0, pn, t, tn, 1.0, noise)
    c0, c1, c2 = OD.ddim_coefficients(buf, t, tn)
    step = DmhStep(objective=ops.OBJECTIVE[objective], clip=1, mode=ops.MODE_DDIM, cond_scale=s,
                   sqrt_recip_ac=float(buf['sqrt_recip_alphas_cumprod'][t]),
                   sqrt_recipm1_ac=float(buf['sqrt_recipm1_alphas_cumprod'][t]),
                   sqrt_ac=float(buf['sqrt_alphas_cumprod'][t]),
                   sqrt_1m_ac=float(buf['sqrt_one_minus_alphas_cumprod'][t]), c0=c0, c1=c1, c2=c2)
    g = lambda v: v.to(dev())
    img, xs, pnd = ops.sampler_step(step, g(cond), g(null), g(x), g(noise), True, True)
    # elementwise fp32 with the reference's op order and no FMA contraction: bit-exact
    assert torch.equal(xs.cpu(), x0), float((xs.cpu() - x0).abs().max())
    assert torch.equal(pnd.cpu(), pn)
    assert torch.equal(img.cpu(), want)


def test_affine_uint8_qsample(ops):
    x = torch.rand((2, 6, 5, 5), generator=torch.Generator().manual_seed(3))
    assert torch.equal(ops.affine(x.to(dev()), 0.5, 0.5).cpu(), (x + 1) * 0.5)
    assert torch.equal(ops.affine(x.to(dev()), 2., -1.).cpu(), x * 2 - 1)
    assert np.array_equal(ops.to_uint8(x.to(dev())).cpu().numpy(), (x.numpy() * 255).astype(np.uint8))
    edge = torch.tensor([0., 1., 0.999999, 1 / 255., 0.5])
    assert np.array_equal(ops.to_uint8(edge.to(dev())).cpu().numpy(), (edge.numpy() * 255).astype(np.uint8))
    y = x.clone()
    y[:, -2:] = y[:, -2:] * 2 - 1
    assert torch.equal(ops.affine_tail_(x.clone().to(dev()), 4, 2., -1.).cpu(), y)
    ca, cb = torch.tensor([0.3, 0.9]), torch.tensor([0.95, 0.43])
    nz = rand((2, 6, 5, 5), 4)
    want = ca[:, None, None, None] * x + cb[:, None, None, None] * nz
    assert torch.equal(ops.q_sample(x.to(dev()), nz.to(dev()), ca.to(dev()), cb.to(dev())).cpu(), want)

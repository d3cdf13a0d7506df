"""-m gpu parity tests of the training pieces built so far: HIP through the C ABI vs torch autograd
of the same op on the CPU in fp64."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import dev, nhwc, nchw, rand

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from dmhomo_amd import ops as _ops
    _ops.lib()
    return _ops


def _rel(name, got, ref):
    rel = ((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    print(f'[parity] {name}: rel_to_max={rel:.3e} ref_absmax={ref.abs().max().item():.3e}')
    return rel


WG_CASES = [  # name, B, H, W, C0, C1, Cout, k, prologue
    ('3x3 64->64 16x16', 2, 16, 16, 64, 0, 64, 3, 0),
    ('3x3 ragged 8->24 19x23', 2, 19, 23, 8, 0, 24, 3, 0),
    ('3x3 concat 64+32->96 20x12', 2, 20, 12, 64, 32, 96, 3, 0),
    ('3x3 prologue 32->48 18x21', 2, 18, 21, 32, 0, 48, 3, 1),
    ('3x3 128->128 33x17', 3, 33, 17, 128, 0, 128, 3, 0),
    ('1x1 64->384 16x16', 2, 16, 16, 64, 0, 384, 1, 0),
    ('1x1 concat 40+24->72 5x33', 3, 5, 33, 40, 24, 72, 1, 0),
]


@pytest.mark.parametrize('case', WG_CASES, ids=[c[0] for c in WG_CASES])
def test_conv_weight_bias_and_data_gradients(ops, case):
    """dW, db (dmh_conv_wgrad) and dX (dmh_conv2d on dy with the flipped, transposed weight) of a stride-1 conv
    against autograd"""
    name, B, H, W, C0, C1, Cout, k, pro = case
    x = rand((B, C0 + C1, H, W), 70)
    w = rand((Cout, C0 + C1, k, k), 71, (1.0 / ((C0 + C1) * k * k)) ** 0.5)
    b = rand((Cout,), 72, 0.1)
    dy = rand((B, Cout, H, W), 73)
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    coef = None
    xin = xd
    if pro:
        a, bb = 1 + 0.3 * rand((B, C0), 74), 0.5 * rand((B, C0), 75)
        coef = torch.stack([a, bb], 1).contiguous().to(dev())
        xin = F.silu(a.double()[:, :, None, None] * xd + bb.double()[:, :, None, None])
    y = F.conv2d(xin, wd, bd, 1, k // 2)
    gx, gw, gb = torch.autograd.grad(y, (xin if pro else xd, wd, bd), dy.double())
    s0 = nhwc(x[:, :C0])
    s1 = nhwc(x[:, C0:]) if C1 else None
    dw, db = ops.conv_wgrad(nhwc(dy), s0, s1, k=k, in_coef=coef)
    assert _rel(name + ' dW', dw, gw) < 2e-6
    assert _rel(name + ' db', db, gb) < 2e-6
    if not pro:
        pd = ops.conv_dgrad_pack(w.to(dev()), C0 + C1)
        dx = nchw(ops.conv2d(pd, nhwc(dy)))
        assert _rel(name + ' dX', dx, gx) < 3e-6


def _ws(w):
    m = w.mean(dim=(1, 2, 3), keepdim=True)
    v = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return (w - m) * (v + 1e-5).rsqrt()


def _ref_block(x, w, b, g, be, ss):
    y = F.group_norm(F.conv2d(x, _ws(w), b, 1, 1), 8, g, be, 1e-5)
    if ss is not None:
        c = y.shape[1]
        y = y * (ss[:, :c, None, None] + 1) + ss[:, c:, None, None]
    return F.silu(y)


@pytest.mark.parametrize('c0,c1,cout,H,W', [(64, 0, 64, 16, 16), (32, 0, 32, 19, 23), (64, 32, 48, 20, 12),
                                            (128, 0, 256, 9, 17)])
def test_resnet_block_backward(ops, c0, c1, cout, H, W):
    """one whole ResnetBlock (CFG:216-241: two weight-standardised convs, two GroupNorms, scale/shift, SiLU, residual
    or 1x1 res_conv, fused torch.cat input) forward + backward on the HIP kernels vs torch autograd in fp64"""
    from dmhomo_amd import train
    B, cin = 2, c0 + c1
    P = dict(w1=rand((cout, cin, 3, 3), 80, (1.0 / (cin * 9)) ** 0.5) + 0.01, b1=rand((cout,), 81, 0.1),
             g1=1 + 0.2 * rand((cout,), 82), be1=0.2 * rand((cout,), 83),
             w2=rand((cout, cout, 3, 3), 84, (1.0 / (cout * 9)) ** 0.5) + 0.01, b2=rand((cout,), 85, 0.1),
             g2=1 + 0.2 * rand((cout,), 86), be2=0.2 * rand((cout,), 87))
    if cin != cout:
        P['rw'], P['rb'] = rand((cout, cin, 1, 1), 88, cin ** -0.5), rand((cout,), 89, 0.1)
    x = rand((B, cin, H, W), 90)
    ss = 0.3 * rand((B, 2 * cout), 91)
    dout = rand((B, cout, H, W), 92)
    # ---- reference: autograd in fp64
    D = {k: v.double().requires_grad_(True) for k, v in P.items()}
    xd, ssd = x.double().requires_grad_(True), ss.double().requires_grad_(True)
    h = _ref_block(xd, D['w1'], D['b1'], D['g1'], D['be1'], ssd)
    h = _ref_block(h, D['w2'], D['b2'], D['g2'], D['be2'], None)
    out = h + (F.conv2d(xd, D['rw'], D['rb']) if 'rw' in D else xd)
    names = list(D)
    grads = torch.autograd.grad(out, [xd, ssd] + [D[k] for k in names], dout.double())
    ref = dict(zip(['x', 'ss'] + names, grads))
    # ---- HIP
    blk = train.ResnetBlockTrain({k: v.to(dev()) for k, v in P.items()}, c0, c1)
    x0 = nhwc(x[:, :c0])
    x1 = nhwc(x[:, c0:]) if c1 else None
    o, saved = blk.forward(x0, x1, ss.to(dev()).contiguous())
    assert _rel('block out', nchw(o), out.detach()) < 2e-5
    dx, g = blk.backward(saved, nhwc(dout))
    tag = f'block {cin}->{cout} {H}x{W}'
    assert _rel(tag + ' dx', nchw(dx), ref['x']) < 5e-5
    assert _rel(tag + ' dss', g['ss'], ref['ss']) < 5e-5
    for k in names:
        assert _rel(tag + ' d' + k, g[k].reshape(ref[k].shape), ref[k]) < 5e-5, k


@pytest.mark.parametrize('C,H,W', [(64, 16, 16), (8, 5, 7), (128, 9, 11), (512, 4, 4)])
def test_chan_layernorm_backward(ops, C, H, W):
    x = rand((2, C, H, W), 100) * 1.5 + 0.3
    g = 1 + 0.2 * rand((C,), 101)
    dout = rand((2, C, H, W), 102)
    xd, gd = x.double().requires_grad_(True), g.double().requires_grad_(True)
    m = xd.mean(1, keepdim=True)
    v = xd.var(1, unbiased=False, keepdim=True)
    out = (xd - m) * (v + 1e-5).rsqrt() * gd[None, :, None, None]
    gx, gg = torch.autograd.grad(out, (xd, gd), dout.double())
    dx, dg = ops.chan_layernorm_backward(nhwc(x), g.to(dev()), nhwc(dout))
    assert _rel(f'LN bwd C={C} dx', nchw(dx), gx) < 5e-6
    assert _rel(f'LN bwd C={C} dg', dg, gg) < 5e-6


@pytest.mark.parametrize('H,W', [(16, 16), (7, 9), (24, 40), (4, 4)])
def test_linear_attention_core_backward(ops, H, W):
    B = 2
    qkv = rand((B, 384, H, W), 110) * 1.5
    dout = rand((B, 128, H, W), 111)
    qd = qkv.double().requires_grad_(True)
    n = H * W
    q, k, v = [t.reshape(B, 4, 32, n) for t in qd.chunk(3, dim=1)]
    q = q.softmax(dim=-2) * 32 ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum('b h d n, b h e n -> b h d e', k, v / n)
    out = torch.einsum('b h d e, b h d n -> b h e n', ctx, q).reshape(B, 128, H, W)
    (gq,) = torch.autograd.grad(out, (qd,), dout.double())
    o, sv = ops.linear_attention_core_train(nhwc(qkv), 32 ** -0.5)
    assert _rel(f'linattn core fwd {H}x{W}', nchw(o), out.detach()) < 1e-5
    dqkv = nchw(ops.linear_attention_core_backward(sv, nhwc(dout)))
    for name, sl in (('dq', slice(0, 128)), ('dk', slice(128, 256)), ('dv', slice(256, 384))):
        assert _rel(f'linattn bwd {H}x{W} {name}', dqkv[:, sl], gq[:, sl]) < 2e-5, name


@pytest.mark.parametrize('H,W', [(16, 16), (7, 9), (4, 4)])
def test_attention_core_forward_backward_small_gemm(ops, H, W):
    B = 2
    qkv = rand((B, 384, H, W), 120) * 1.5
    dout = rand((B, 128, H, W), 121)
    qd = qkv.double().requires_grad_(True)
    n = H * W
    q, k, v = [t.reshape(B, 4, 32, n) for t in qd.chunk(3, dim=1)]
    sim = torch.einsum('b h d i, b h d j -> b h i j', q * 32 ** -0.5, k)
    out = torch.einsum('b h i j, b h d j -> b h i d', sim.softmax(dim=-1), v).permute(0, 1, 3, 2).reshape(B, 128, H, W)
    (gq,) = torch.autograd.grad(out, (qd,), dout.double())
    o, sv = ops.attention_core_train(nhwc(qkv), 32 ** -0.5)
    assert _rel(f'attn fwd {H}x{W}', nchw(o), out.detach()) < 1e-5
    dqkv = nchw(ops.attention_core_backward(sv, nhwc(dout)))
    for name, sl in (('dq', slice(0, 128)), ('dk', slice(128, 256)), ('dv', slice(256, 384))):
        assert _rel(f'attn bwd {H}x{W} {name}', dqkv[:, sl], gq[:, sl]) < 2e-5, name


def test_init_conv_7x7_weight_gradient(ops):
    B, H, W, C, Co = 2, 21, 18, 12, 64
    x, w, b, dy = rand((B, C, H, W), 130), rand((Co, C, 7, 7), 131, 0.05), rand((Co,), 132, 0.1), rand((B, Co, H, W), 133)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    gw, gb = torch.autograd.grad(F.conv2d(x.double(), wd, bd, 1, 3), (wd, bd), dy.double())
    dw, db = ops.conv_wgrad(nhwc(dy), nhwc(x), k=7)
    assert _rel('7x7 dW', dw, gw) < 2e-6 and _rel('7x7 db', db, gb) < 2e-6


@pytest.mark.parametrize('C,Co,H,W', [(64, 64, 16, 16), (32, 48, 12, 20), (128, 256, 8, 8)])
def test_downsample_conv_backward(ops, C, Co, H, W):
    """Downsample (conv 4x4 / stride 2 / pad 1, CFG:110-111): dX through a 3x3 conv over dy + pixel shuffle, dW through the
    2x2 form over the shifted space-to-depth view"""
    B = 2
    x, w, b = rand((B, C, H, W), 140), rand((Co, C, 4, 4), 141, (1.0 / (C * 16)) ** 0.5), rand((Co,), 142, 0.1)
    dy = rand((B, Co, H // 2, W // 2), 143)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    gx, gw, gb = torch.autograd.grad(F.conv2d(xd, wd, bd, 2, 1), (xd, wd, bd), dy.double())
    dx, dw, db = ops.conv_down_backward(nhwc(dy), nhwc(x), w.to(dev()))
    assert _rel('down dX', nchw(dx), gx) < 3e-6
    assert _rel('down dW', dw, gw) < 3e-6
    assert _rel('down db', db, gb) < 3e-6


@pytest.mark.parametrize('C,Co,H,W', [(128, 64, 8, 8), (32, 16, 7, 9)])
def test_upsample_conv_backward(ops, C, Co, H, W):
    """Upsample (nearest x2 -> conv3x3, CFG:106-107)"""
    B = 2
    x, w, b = rand((B, C, H, W), 150), rand((Co, C, 3, 3), 151, (1.0 / (C * 9)) ** 0.5), rand((Co,), 152, 0.1)
    dy = rand((B, Co, 2 * H, 2 * W), 153)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y = F.conv2d(F.interpolate(xd, scale_factor=2, mode='nearest'), wd, bd, 1, 1)
    gx, gw, gb = torch.autograd.grad(y, (xd, wd, bd), dy.double())
    dx, dw, db = ops.conv_up_backward(nhwc(dy), nhwc(x), w.to(dev()))
    assert _rel('up dX', nchw(dx), gx) < 3e-6
    assert _rel('up dW', dw, gw) < 3e-6
    assert _rel('up db', db, gb) < 3e-6


def test_unet_backward_full_width_ragged_vs_autograd():
    """the same at the DGM width (dim 64: 64..512 channels, every fp16-piece weight-gradient / data-gradient shape of the
    real model, the sub-pixel upsampling convs in the training forward) on a 24x24 image (ragged tiles at every level)"""
    from test_gpu_unet import make_cfg, _cond_inputs, g
    from dmhomo_amd import train
    from oracle import unet as OU
    m, sd = make_cfg(64)
    B, S = 2, 24
    x, rf, mk = _cond_inputs(B, S, 710)
    t = torch.tensor([17, 803])
    c = torch.zeros(B, dtype=torch.long)
    keep = torch.tensor([True, False])
    dout = rand((B, 6, S, S), 711)
    sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    out_ref = OU.cfg_unet_forward(sdd, x, t, c, rf, mk, keep)
    pnames = [k for k, _ in m.named_parameters()]
    ref = dict(zip(pnames, torch.autograd.grad(out_ref, [sdd[k] for k in pnames], dout, allow_unused=True)))
    ut = train.UnetTrain(m)
    out, saved = ut.forward(g(x), g(t), g(c), g(rf), g(mk), g(keep))
    assert _rel('unet64 train fwd', out, out_ref.detach()) < 2e-4
    got = ut.backward(saved, g(dout))
    worst = 0.0
    for k in pnames:
        if ref[k] is None:
            continue
        if ref[k].abs().max() < 1e-4:
            assert got[k].abs().max().item() < 1e-3, k
            continue
        r = ((got[k].double().cpu().reshape(ref[k].shape) - ref[k].double()).abs().max() /
             ref[k].double().abs().max().clamp_min(1e-30)).item()
        if r > 1e-4:
            print(f'[parity] unet64 bwd {k}: rel_to_max={r:.3e}')
        worst = max(worst, r)
    print(f'[parity] unet64 bwd (24x24): {len(pnames)} parameter gradients, worst rel_to_max={worst:.3e}')
    assert worst < 6e-5, worst          # measured 5.8e-6


def test_unet_backward_config3_geometry_vs_autograd():
    """BASELINE configs[3]'s own geometry (dim 64, 128x128; B = 2 of the 16 images a GPU holds): the HIP backward
    (CFG:770-842 -> loss.backward(), DDP:1843-1850) against torch autograd through the oracle's forward — every parameter,
    with one gradient per kernel family named in the log: the 7x7 init conv, a weight-standardised 3x3 conv at 128x128,
    a LinearAttention to_qkv, a GroupNorm gain, the bottleneck attention's to_out, the final 1x1 conv"""
    import os
    from test_gpu_unet import make_cfg, _cond_inputs, g
    from dmhomo_amd import train
    from oracle import unet as OU
    m, sd = make_cfg(64)
    B, S = 2, 128
    x, rf, mk = _cond_inputs(B, S, 730)
    t = torch.tensor([17, 803])
    c = torch.zeros(B, dtype=torch.long)
    keep = torch.tensor([True, False])
    dout = rand((B, 6, S, S), 731) * (1.0 / (6 * S * S))          # the scale of d(mean loss)/d(out)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    out_ref = OU.cfg_unet_forward(sdd, x, t, c, rf, mk, keep)
    pnames = [k for k, _ in m.named_parameters()]
    ref = dict(zip(pnames, torch.autograd.grad(out_ref, [sdd[k] for k in pnames], dout, allow_unused=True)))
    ut = train.UnetTrain(m)
    out, saved = ut.forward(g(x), g(t), g(c), g(rf), g(mk), g(keep))
    assert _rel('unet64 @128 train fwd', out, out_ref.detach()) < 2e-4
    got = ut.backward(saved, g(dout))
    named = ('init_conv.weight', 'downs.0.0.block1.proj.weight', 'downs.0.2.fn.fn.to_qkv.weight',
             'downs.0.1.block2.norm.weight', 'mid_attn.fn.fn.to_out.weight', 'final_conv.weight')
    worst, errs = 0.0, {}
    for k in pnames:
        if ref[k] is None:
            continue
        scale = ref[k].double().abs().max().clamp_min(1e-30)
        r = ((got[k].double().cpu().reshape(ref[k].shape) - ref[k].double()).abs().max() / scale).item()
        errs[k] = r
        worst = max(worst, r)
    for k in named:
        print(f'[parity] unet64 @128x128 bwd {k}: rel_to_max={errs[k]:.3e}')
    wk = max(errs, key=errs.get)
    print(f'[parity] unet64 @128x128 bwd: {len(errs)} parameter gradients, worst rel_to_max={worst:.3e} ({wk})')
    assert worst < 1e-4, (worst, wk)


def test_unet_backward_vs_autograd():
    """the whole conditional UNet (CFG:412-466), tiny geometry: forward with saved activations + backward on the HIP
    kernels against torch autograd through the oracle's functional forward in fp64 — every parameter's gradient"""
    from test_gpu_unet import make_cfg, _cond_inputs, g
    from dmhomo_amd import train
    from oracle import unet as OU
    m, sd = make_cfg(8)
    B, S = 3, 16
    x, rf, mk = _cond_inputs(B, S, 700)
    t = torch.tensor([17, 503, 998])
    c = torch.zeros(B, dtype=torch.long)
    keep = torch.tensor([True, False, True])
    dout = rand((B, 6, S, S), 701)
    # ---- reference
    # (fp32 on purpose: the reference's weight standardisation switches to eps = 1e-3 for any other dtype, CFG:121)
    sdd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    out_ref = OU.cfg_unet_forward(sdd, x, t, c, rf, mk, keep)
    pnames = [k for k, _ in m.named_parameters()]
    grads = torch.autograd.grad(out_ref, [sdd[k] for k in pnames], dout, allow_unused=True)
    ref = dict(zip(pnames, grads))
    # ---- HIP
    ut = train.UnetTrain(m)
    out, saved = ut.forward(g(x), g(t), g(c), g(rf), g(mk), g(keep))
    assert _rel('unet train fwd', out, out_ref.detach()) < 2e-4
    got = ut.backward(saved, g(dout))
    worst, missing = 0.0, [k for k in pnames if k not in got and ref[k] is not None]
    assert not missing, missing
    for k in pnames:
        if ref[k] is None:
            continue
        if ref[k].abs().max() < 1e-4:            # e.g. a conv bias in front of a GroupNorm: its true gradient is 0
            assert got[k].abs().max().item() < 1e-3, k
            continue
        r = ((got[k].double().cpu().reshape(ref[k].shape) - ref[k].double()).abs().max() /
             ref[k].double().abs().max().clamp_min(1e-30)).item()
        if r > 1e-4:
            print(f'[parity] unet bwd {k}: rel_to_max={r:.3e} ref_absmax={ref[k].abs().max().item():.3e}')
        worst = max(worst, r)
    print(f'[parity] unet bwd: {len(pnames)} parameter gradients, worst rel_to_max={worst:.3e}')
    assert worst < 6e-5, worst          # measured 5.8e-6


# ----------------------------------------------------------------------------------------------- training step
def _ref_flow_warp(x, flow):
    """the reference's flow_warp (DDP:1262-1280) as the torch op it calls: grid_sample, bilinear, border, align_corners"""
    B, _, H, W = x.shape
    xs = torch.arange(W, dtype=x.dtype).view(1, 1, W).expand(B, H, W)
    ys = torch.arange(H, dtype=x.dtype).view(1, H, 1).expand(B, H, W)
    gx = 2.0 * (xs + flow[:, 0]) / (W - 1) - 1.0
    gy = 2.0 * (ys + flow[:, 1]) / (H - 1) - 1.0
    return F.grid_sample(x, torch.stack([gx, gy], -1), mode='bilinear', padding_mode='border', align_corners=True)


@pytest.mark.parametrize('squared', [False, True], ids=['l1', 'l2'])
def test_loss_backward_vs_autograd(ops, squared):
    """gradient of p_losses (CFG:796-806) wrt the UNet output, incl. the transpose of grid_sample; flows leave the
    image on every side (border clamp) and fold over themselves"""
    B, H, W = 3, 24, 40
    out = rand((B, 6, H, W), 800)
    target = rand((B, 6, H, W), 801)
    mask = (torch.rand((B, 1, H, W), generator=torch.Generator().manual_seed(802)) > 0.4).float()
    flow = rand((B, 2, H, W), 803, 6.0)
    abar = torch.tensor([0.9, 0.31, 0.004])
    od = out.double().requires_grad_(True)
    fn = F.mse_loss if squared else F.l1_loss
    warped_ref = _ref_flow_warp(od[:, 3:], flow.double())
    loss = fn(od, target.double(), reduction='none').reshape(B, -1).mean() + \
        (abar.double()[:, None] * (mask.double() * fn(warped_ref, od[:, :3], reduction='none')).reshape(B, -1)).mean()
    gref, = torch.autograd.grad(loss, od)
    o = out.to(dev())
    warped = ops.flow_warp(o[:, 3:].contiguous(), flow.to(dev()))
    got = ops.loss_backward(o, target.to(dev()), warped, mask.to(dev()), flow.to(dev()), abar.to(dev()), squared)
    assert _rel('loss backward', got, gref) < 5e-6


def test_adam_clip_ema_vs_torch(ops):
    """dmh_sumsq / dmh_gradnorm_finalize / dmh_adam / dmh_ema against clip_grad_norm_ + torch.optim.Adam + lerp_"""
    shapes = [(64, 64, 3, 3), (257,), (1, 8, 1, 1), (33, 129)]
    ps = [rand(s, 810 + i) for i, s in enumerate(shapes)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-3, betas=(0.9, 0.99))
    mine = [p.clone().to(dev()) for p in ps]
    m = [torch.zeros_like(p) for p in mine]
    v = [torch.zeros_like(p) for p in mine]
    ema_ref = [p.clone() for p in ps]
    ema = [p.clone().to(dev()) for p in ps]
    for step in range(1, 5):
        gs = [rand(s, 820 + 10 * step + i, 0.02 * step) for i, s in enumerate(shapes)]
        for r, gr in zip(ref, gs):
            r.grad = gr.clone()
        nrm = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        opt.step()
        clip = ops.grad_norm_clip([gr.to(dev()) for gr in gs], 1.0)
        assert abs(clip[0].item() - nrm.item()) <= 1e-6 * nrm.item()
        for p_, gr, m_, v_ in zip(mine, gs, m, v):
            ops.adam_(p_, gr.to(dev()), m_, v_, clip, 1e-3, 0.9, 0.99, 1e-8, step)
        for e_, er, p_, r in zip(ema, ema_ref, mine, ref):
            ops.ema_(e_, p_, 0.9)
            er.lerp_(r.detach(), 1 - 0.9)
    for i, (p_, r) in enumerate(zip(mine, ref)):
        d = (p_.cpu() - r.detach()).abs().max().item()
        print(f'[parity] adam tensor {i}: max abs diff {d:.3e} after 4 steps of ~1e-3')
        assert d < 5e-7                           # parameters reach |4|: one fp32 ulp there is 4.8e-7
        assert (ema[i].cpu() - ema_ref[i]).abs().max().item() < 5e-7


def _train_setup(golden_dir, lr=1e-3, accum=1):
    import numpy as np
    import os
    from test_gpu_unet import make_cfg, g
    from dmhomo_amd import cfg, train
    gd = {k: v for k, v in np.load(os.path.join(golden_dir, 'train_step.npz')).items()}
    m, sd = make_cfg(8)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective='pred_x0',
                              loss_type='l1').to(dev())
    ts = train.TrainStep(d, lr=lr, betas=(0.9, 0.99), accum=accum)
    T = lambda k: torch.from_numpy(gd[k])
    draws = dict(t=g(T('t')), noise=g(T('noise')), keep=g(T('keep')))
    return gd, m, d, ts, g(T('img12')), g(T('classes')), draws


def test_train_step_gradients_vs_reference(golden_dir):
    """the reference's own loss.backward() (tests/golden/make_golden_train.py): loss, every parameter gradient, and the
    global norm clip_grad_norm_ reports"""
    gd, m, d, ts, img, classes, draws = _train_setup(golden_dir)
    loss, grads = ts.loss_and_grads(img, classes, **draws)
    want = float(gd['loss'])
    print(f'[parity] train loss: got {float(loss):.7f} want {want:.7f}')
    assert abs(float(loss) - want) <= 2e-5 * abs(want)
    worst = 0.0
    for k, _ in m.named_parameters():
        ref = torch.from_numpy(gd['grad.' + k]).double()
        got = grads[k].double().cpu().reshape(ref.shape)
        if ref.abs().max() < 1e-5:                # conv biases in front of a GroupNorm: true gradient 0
            assert got.abs().max().item() < 1e-4, k
            continue
        r = ((got - ref).abs().max() / ref.abs().max()).item()
        if r > 1e-4:
            print(f'[parity] train grad {k}: rel_to_max={r:.3e}')
        worst = max(worst, r)
    clip = ts.apply(grads)
    print(f'[parity] train step: worst gradient rel_to_max={worst:.3e}; grad norm {clip[0].item():.6f} '
          f'want {float(gd["grad_norm"]):.6f}')
    assert worst < 6e-5, worst           # measured 5.8e-6
    assert abs(clip[0].item() - float(gd['grad_norm'])) <= 1e-4 * float(gd['grad_norm'])


def test_train_trajectory_vs_reference(golden_dir):
    """6 optimiser steps of DDP:1840-1858 (accumulate 2, clip 1.0, Adam lr 1e-3) on a fixed batch with fixed draws: the
    loss the reference's own loop printed at each step, and the parameter norm after it"""
    gd, m, d, ts, img, classes, draws = _train_setup(golden_dir, lr=1e-3, accum=2)
    for i in range(6):
        total = ts.step([(img, classes), (img, classes)], draws=[draws, draws])
        pl2 = float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in m.parameters())))
        want, wl2 = float(gd['traj.loss'][i]), float(gd['traj.param_l2'][i])
        print(f'[parity] train step {i}: loss {float(total):.6f} want {want:.6f}   |params| {pl2:.6f} want {wl2:.6f}')
        assert abs(float(total) - want) <= (1e-5 if i == 0 else 1e-4) * want
        assert abs(pl2 - wl2) <= 1e-4 * wl2
    # the sampling engine sees the trained weights (its packed copies are rebuilt on the weight epoch)
    out = m(draws['noise'], draws['t'], classes, img[:, -5:-2].contiguous(), img[:, 6:7].contiguous(), cond_drop_prob=0.)
    out2, _ = ts.ut.forward(draws['noise'], draws['t'], classes, img[:, -5:-2].contiguous(), img[:, 6:7].contiguous(),
                            torch.ones(3, dtype=torch.uint8, device=img.device))
    assert (out - out2).abs().max().item() < 2e-4 * out2.abs().max().item()


def test_train_step_follows_moved_parameter_storage(golden_dir):
    """a parameter whose storage moves between two optimiser steps (p.data = clone(), what .to() / .float() /
    load-by-assignment do) must be re-packed from its NEW storage: the re-pack HIP graph captured on the first step
    holds the old pointers and may not be replayed (the old storage is overwritten with NaN here to make a stale read
    visible).  The trajectory must equal the one of an undisturbed twin (to fp32 rounding: the batch-reduction
    kernels of the backward pass are not bitwise run-to-run reproducible)."""
    gd, m, d, ts, img, classes, draws = _train_setup(golden_dir, lr=1e-3, accum=1)
    gd2, m2, d2, ts2, img2, classes2, draws2 = _train_setup(golden_dir, lr=1e-3, accum=1)
    a1, b1 = ts.step([(img, classes)], draws=[draws]), ts2.step([(img2, classes2)], draws=[draws2])
    assert abs(float(a1) - float(b1)) <= 1e-6 * abs(float(b1))
    old = []
    with torch.no_grad():
        for p in m.parameters():
            stale = p.data
            p.data = stale.clone()
            old.append(stale)
        for stale in old:
            stale.fill_(float('nan'))
    for i in range(2):
        a, b = ts.step([(img, classes)], draws=[draws]), ts2.step([(img2, classes2)], draws=[draws2])
        assert abs(float(a) - float(b)) <= 2e-6 * abs(float(b)), (i, float(a), float(b))
    for (k, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.isfinite(p.detach()).all(), k                                  # a stale read would be NaN
        if k.endswith('.proj.bias'):
            continue      # a conv bias in front of a GroupNorm: its true gradient is 0, Adam amplifies the rounding noise
        # Adam divides by sqrt(v): an element whose gradient sits at the rounding floor moves by up to lr per step in
        # either twin, so the element-wise bound is 3 steps x lr and the closeness is judged on the L2 norm
        diff = (p.detach() - q.detach()).double()
        assert diff.abs().max().item() <= 3.05e-3, (k, diff.abs().max().item())
        assert diff.norm().item() <= 1e-4 * max(1e-3, q.detach().double().norm().item()), (k, diff.norm().item())
    # the sampling engine follows too
    x = draws['noise']
    o1 = m(x, draws['t'], classes, img[:, -5:-2].contiguous(), img[:, 6:7].contiguous(), cond_drop_prob=0.)
    o2 = m2(x, draws['t'], classes, img[:, -5:-2].contiguous(), img[:, 6:7].contiguous(), cond_drop_prob=0.)
    assert torch.isfinite(o1).all() and (o1 - o2).abs().max().item() <= 1e-4 * o2.abs().max().item()


def test_trainer_train_loop_and_checkpoint(golden_dir, tmp_path):
    """Trainer.train (DDP:1828-1940): steps, EMA schedule, checkpoint with optimiser state, resume"""
    from test_gpu_unet import make_cfg
    from dmhomo_amd import cfg, ddpm
    m, _ = make_cfg(8)
    d = cfg.GaussianDiffusion(m, image_size=32, timesteps=1000, sampling_timesteps=2, objective='pred_x0').to(dev())
    w0 = m.init_conv.weight.detach().clone()
    tr = ddpm.Trainer(d, 'DGM_Conditions', train_batch_size=2, gradient_accumulate_every=2, train_lr=1e-3,
                      train_num_steps=4, results_folder=str(tmp_path), save_and_sample_every=4, ema_update_every=1)
    tr.ema.update_after_step = 1
    losses = []
    tr.train(log=lambda s, l: losses.append(float(l)))
    assert tr.step == 4 and len(losses) == 4 and all(l == l for l in losses)
    assert not torch.equal(w0, m.init_conv.weight.detach())
    assert tr.ema.ema_model is not tr.ema.online_model and int(tr.ema.step) == 4
    ck = torch.load(str(tmp_path / 'model-1.pt'), map_location='cpu')
    assert set(ck) == {'step', 'model', 'opt', 'ema', 'scaler', 'version'} and ck['step'] == 4
    assert len(ck['opt']['state']) == len(list(d.parameters())) and ck['opt']['param_groups'][0]['lr'] == 1e-3
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().cpu().clone()) for p in d.parameters()], lr=1.0)
    ref.load_state_dict(ck['opt'])                      # the layout torch.optim.Adam itself accepts
    # EMA with update_after_step = 1: copies on updates 1-3, a lerp with decay 1 - 3^(-2/3) on the 4th
    e, o = tr.ema.ema_model.model.init_conv.weight, m.init_conv.weight
    assert not torch.equal(e.detach(), o.detach())
    m2, _ = make_cfg(8, seed=5)
    d2 = cfg.GaussianDiffusion(m2, image_size=32, timesteps=1000, sampling_timesteps=2, objective='pred_x0').to(dev())
    tr2 = ddpm.Trainer(d2, 'DGM_Conditions', train_batch_size=2, gradient_accumulate_every=2, train_lr=1e-3,
                       train_num_steps=5, results_folder=str(tmp_path))
    tr2.load(1)
    assert tr2.step == 4 and torch.equal(m2.init_conv.weight.detach(), m.init_conv.weight.detach())
    assert torch.equal(tr2.ema.ema_model.model.init_conv.weight.detach(), e.detach())
    ts2 = tr2.train_step_engine()
    assert ts2.opt_step == 4 and torch.equal(ts2.m['init_conv.weight'], tr.train_step_engine().m['init_conv.weight'])
    tr2.train()
    assert tr2.step == 5


def test_user_loop_loss_backward_and_torch_optimizer(golden_dir):
    """the reference's own loop shape (DDP:1843-1857) written by a user: loss = diffusion(batch, classes=c);
    loss.backward(); clip_grad_norm_; torch.optim.Adam.step() — the loss carries a grad_fn whose gradients are the HIP
    backward kernels'; an external optimiser's in-place update is noticed (weights re-packed) on the next call"""
    gd, m, d, ts, img, classes, draws = _train_setup(golden_dir)
    opt = torch.optim.Adam(d.parameters(), lr=1e-3, betas=(0.9, 0.99))

    class Fixed:                                   # replay the golden draws through the module's own RNG hooks
        def randn(self, shape, device):
            return draws['noise']

        def uniform(self, n, device):
            return torch.where(draws['keep'].bool(), 0.25, 0.75)

    d.rng = m.rng = Fixed()
    want_t = draws['t']
    orig = torch.randint
    torch.randint = lambda *a, **k: want_t          # CFG:812 draws t with torch.randint
    try:
        losses = []
        for i in range(3):
            loss = d(img, classes=classes)
            assert loss.requires_grad and loss.grad_fn is not None
            loss.backward()
            if i == 0:
                ref = torch.from_numpy(gd['grad.init_conv.weight'])
                got = m.init_conv.weight.grad.cpu()
                assert ((got - ref).abs().max() / ref.abs().max()).item() < 1e-4
                assert abs(float(loss.detach()) - float(gd['loss'])) < 2e-5 * float(gd['loss'])
            torch.nn.utils.clip_grad_norm_(d.parameters(), 1.0)
            opt.step()
            opt.zero_grad()
            losses.append(float(loss.detach()))
        with torch.no_grad():
            val = d(img, classes=classes)
        assert not val.requires_grad
    finally:
        torch.randint = orig
    print('[parity] user loop losses', losses)
    assert losses[2] < losses[1] < losses[0]


def test_conv_wgrad_dynamic_range(ops):
    """the fp16-piece weight-gradient kernel under the conditions its block scaling exists for: gradient magnitudes
    from 1e-9 to 1e+3 and activations from 1e-4 to 1e+4 varying over decades ACROSS tiles (the running maxima rise and
    fall along a workgroup's item range, the accumulators get rescaled), whole tiles of exact zeros (masked loss), and a
    few isolated huge values — against fp64 autograd, error measured against the result's own scale"""
    B, H, W, C, Co = 3, 32, 48, 64, 64
    g = torch.Generator().manual_seed(910)
    dy = torch.randn((B, Co, H, W), generator=g)
    x = torch.randn((B, C, H, W), generator=g)
    ramp = torch.logspace(-9, 3, W).view(1, 1, 1, W) * torch.logspace(0, -3, H).view(1, 1, H, 1)
    dy = dy * ramp * torch.tensor([1.0, 1e-3, 1e2]).view(B, 1, 1, 1)
    dy[:, :, 8:16, :] = 0.0                                         # zero tiles
    dy[1, 5, 20, 7] = 3.0e3                                         # outliers
    x = x * torch.logspace(4, -4, H).view(1, 1, H, 1)
    x[2, 9, 3, 40] = -7.0e4
    wd = torch.zeros((Co, C, 3, 3), dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), wd, None, 1, 1)
    gw, = torch.autograd.grad(y, wd, dy.double())
    dw, db = ops.conv_wgrad(nhwc(dy), nhwc(x), None, k=3)
    assert _rel('wgrad dynamic range dW', dw, gw) < 2e-6
    assert _rel('wgrad dynamic range db', db, dy.double().sum((0, 2, 3))) < 2e-6
    # per-tap check against each tap's own scale (a tap dominated by small-magnitude tiles must not drown)
    for ky in range(3):
        for kx in range(3):
            r = ((dw[:, :, ky, kx].double().cpu() - gw[:, :, ky, kx]).abs().max() / gw[:, :, ky, kx].abs().max()).item()
            assert r < 5e-6, (ky, kx, r)


@pytest.mark.parametrize('variant', ['0', '6'])
def test_training_under_exact_fp32_conv_variants_in_child_process(variant):
    """DMH_CONV3_VARIANT != 9 (the exact-fp32 forward kernels; read once per process): the training step builds — the
    table-driven weight re-pack only makes the fp16-piece images, so the standardised weights and the data-gradient
    weights go through fixed buffers + per-weight pack launches — and follows the reference's gradients and its 6-step
    optimiser trajectory (which needs every image re-made after every update)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMH_CONV3_VARIANT=variant)
    here = os.path.join(root, 'tests', 'test_gpu_backward.py')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        here + '::test_train_step_gradients_vs_reference', here + '::test_train_trajectory_vs_reference'],
                       capture_output=True, text=True, env=env, cwd=root, timeout=1500)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]

"""-m gpu: the UNets and samplers end to end (HIP path through the reference-shaped classes) against
the CPU oracle and the golden vectors captured from the reference.

Tolerances: <= 10x the error measured on MI355X (fp32, different accumulation order than oneDNN; the measured
numbers are printed as [parity] lines and recorded in DESIGN.md §4), relative to each tensor's own scale:
  per-layer activations       2e-5 of the layer's max |x|     (measured 1.1e-6 .. 3.1e-6)
  one UNet forward, output    2e-5 of max |out|               (measured 1.9e-6)
  S-step sampler output       4e-4 abs on the [0,1] image     (measured 1.3e-5 at S=4 tiny, 3.7e-5 at S=4 full size),
                              uint8 export within +-1 LSB
"""
import json
import os

import numpy as np
import pytest
import torch

from gpu_util import dev, nchw, close, close_rel, report, rand, ReplayDeviceRng
from detweights import det_state_dict, shapes_of
from oracle import unet as OU, diffusion as OD

pytestmark = pytest.mark.gpu

LAYER_REL = 2e-5        # per-layer activation error / max |activation|   (measured <= 3.1e-6)
OUT_REL = 2e-5          # UNet output error / max |output|                (measured 1.9e-6)


def load(golden_dir, name):
    return {k: v for k, v in np.load(os.path.join(golden_dir, name + '.npz')).items()}


def T(a):
    return torch.from_numpy(np.asarray(a))


def make_cfg(dim, seed=0, out_scale=1.0, **kw):
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1, **kw)
    sd = det_state_dict(shapes_of(m), seed, out_scale)
    m.load_state_dict(sd)
    return m.to(dev()), sd


def make_ddp(dim, sc, seed=1):
    from dmhomo_amd import ddpm
    m = ddpm.Unet(dim=dim, dim_mults=(1, 2, 4, 8), channels=3, self_condition=sc)
    sd = det_state_dict(shapes_of(m), seed)
    m.load_state_dict(sd)
    return m.to(dev()), sd


def g(x):
    return x.to(dev())


# ------------------------------------------------------------------------------- UNet forward
def test_unet_cfg_tiny_vs_golden(golden_dir):
    """F1: reference outputs (dim=8, 32x32) incl. per-layer taps"""
    gd = load(golden_dir, 'unet_cfg_tiny')
    m, sd = make_cfg(8)
    x, t, c, rf, mk = (T(gd[k]) for k in ('x', 't', 'classes', 'rgb_flow', 'mask'))
    taps = {}
    out = m._run(g(x), g(t), g(c), g(rf), g(mk), [None], taps=taps)
    worst = 0.
    for k, v in taps.items():
        if 'tap.' + k in gd:
            a, r = report('tap ' + k, nchw(v), T(gd['tap.' + k]))
            worst = max(worst, r)
    assert worst < LAYER_REL, worst
    close_rel('cfg tiny cond', out.cpu(), T(gd['out_cond']), OUT_REL)
    out = m(g(x), g(t), g(c), g(rf), g(mk), cond_drop_prob=1.)
    close_rel('cfg tiny null', out.cpu(), T(gd['out_null']), OUT_REL)
    m.rng = ReplayDeviceRng([torch.where(T(gd['keep_half']), 0.25, 0.75)])     # uniform draw reproducing keep
    out = m(g(x), g(t), g(c), g(rf), g(mk))
    close_rel('cfg tiny half', out.cpu(), T(gd['out_half']), OUT_REL)
    m.rng = ReplayDeviceRng([torch.where(T(gd['keep_scale3']), 0.25, 0.75)])
    out = m.forward_with_cond_scale(g(x), g(t), g(c), rgb_flow=g(rf), mask=g(mk), cond_scale=3.)
    close_rel('cfg tiny scale3', out.cpu(), T(gd['out_scale3']), 3 * OUT_REL)     # 3*cond - 2*null: errors add up x5


@pytest.mark.parametrize('tag', ['nosc', 'sc'])
def test_unet_ddp_tiny_vs_golden(golden_dir, tag):
    gd = load(golden_dir, 'unet_ddp_tiny')
    m, sd = make_ddp(8, tag == 'sc')
    x, xs, t = T(gd['x']), T(gd['x_self_cond']), T(gd['t'])
    if tag == 'sc':
        close_rel('ddp sc', m(g(x), g(t), g(xs)).cpu(), T(gd['sc.out']), OUT_REL)
        close_rel('ddp sc default', m(g(x), g(t)).cpu(), T(gd['sc.out_default']), OUT_REL)
    else:
        close_rel('ddp nosc', m(g(x), g(t)).cpu(), T(gd['nosc.out']), OUT_REL)


def _cond_inputs(B, S, seed):
    x = rand((B, 6, S, S), seed)
    rf = torch.rand((B, 3, S, S), generator=torch.Generator().manual_seed(seed + 1)) * 2 - 1
    mk = (torch.rand((B, 1, S, S), generator=torch.Generator().manual_seed(seed + 2)) > 0.4).float()
    return x, rf, mk


def test_unet_cfg_fullsize_vs_oracle():
    """BASELINE config geometry: dim=64, 128x128, B=2, per-layer taps against the oracle"""
    m, sd = make_cfg(64)
    x, rf, mk = _cond_inputs(2, 128, 100)
    t = torch.tensor([967, 30])
    c = torch.zeros(2, dtype=torch.long)
    keep = torch.tensor([True, False])
    rtaps, taps = {}, {}
    with torch.no_grad():
        ref = OU.cfg_unet_forward(sd, x, t, c, rf, mk, keep, taps=rtaps)
    out = m._run(g(x), g(t), g(c), g(rf), g(mk), [g(keep.to(torch.uint8))], taps=taps)
    worst = 0.
    for k in rtaps:
        a, r = report('full ' + k, nchw(taps[k]), rtaps[k])
        worst = max(worst, r)
    assert worst < LAYER_REL, worst
    close_rel('cfg full out', out.cpu(), ref, OUT_REL)


@pytest.mark.parametrize('size,B', [(40, 3), (72, 1)])
def test_unet_cfg_ragged_sizes_vs_oracle(size, B):
    """image sizes that are multiples of 8 but not of the kernels' tiles (40: 16-pixel tiles end ragged at every level,
    the 5x5 bottleneck attention has 25 keys, the sub-pixel upsampling convs see 5 / 10 / 20-pixel rows; 72: 9x9
    bottleneck), dim = 64 so that every fused / fp16-piece path is taken; single sample too"""
    m, sd = make_cfg(64)
    x, rf, mk = _cond_inputs(B, size, 300 + size)
    t = torch.tensor([967, 30, 400][:B])
    c = torch.zeros(B, dtype=torch.long)
    keep = torch.tensor([True, False, True][:B])
    rtaps, taps = {}, {}
    with torch.no_grad():
        ref = OU.cfg_unet_forward(sd, x, t, c, rf, mk, keep, taps=rtaps)
    out = m._run(g(x), g(t), g(c), g(rf), g(mk), [g(keep.to(torch.uint8))], taps=taps)
    worst = max(report(f'ragged{size} ' + k, nchw(taps[k]), rtaps[k])[1] for k in rtaps)
    assert worst < LAYER_REL, worst
    close_rel(f'cfg ragged {size} out', out.cpu(), ref, OUT_REL)


@pytest.mark.parametrize('kw', [
    dict(dim=16, dim_mults=(1, 2), channels=3, num_classes=4, resnet_block_groups=4),
    dict(dim=16, dim_mults=(1, 2, 4, 8), channels=6, num_classes=2, out_dim=5),
    dict(dim=24, dim_mults=(1, 3), channels=6, num_classes=1, resnet_block_groups=2),
    dict(dim=8, dim_mults=(1,), channels=6, num_classes=1),
    dict(dim=16, dim_mults=(1, 2), channels=6, num_classes=1, learned_variance=True),
], ids=lambda kw: ','.join(f'{k}={v}' for k, v in kw.items() if k != 'dim_mults') + f",mults={len(kw['dim_mults'])}")
def test_unet_cfg_constructor_variants_vs_oracle(kw):
    """the constructor arguments of CFG:304-318 away from the DGM's values: other widths / depths (one to four stages, a
    non-power-of-two multiplier), 3 image channels, several classes with a mixed keep mask, 2 or 4 GroupNorm groups, an
    explicit out_dim, learned_variance (out_dim = 2 * channels) — forward against the oracle"""
    from dmhomo_amd import cfg
    m = cfg.Unet(**kw)
    sd = det_state_dict(shapes_of(m), 5)
    m.load_state_dict(sd)
    m = m.to(dev())
    B, S = 2, 32
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B, kw['channels'], S, S, generator=gen)
    rf = torch.rand(B, 3, S, S, generator=gen) * 2 - 1
    mk = (torch.rand(B, 1, S, S, generator=gen) > 0.4).float()
    t, c = torch.tensor([700, 20]), torch.randint(0, kw['num_classes'], (B,), generator=gen)
    keep = torch.tensor([True, False])
    with torch.no_grad():
        ref = OU.cfg_unet_forward(sd, x, t, c, rf, mk, keep, groups=kw.get('resnet_block_groups', 8))
    m.rng = ReplayDeviceRng([torch.where(keep, 0.25, 0.75)])
    out = m(g(x), g(t), g(c), g(rf), g(mk))
    assert out.shape == ref.shape
    close_rel(f'CFG Unet {kw}', out, ref, OUT_REL)


def test_unet_rows_are_independent():
    """size-independent property: each output row depends on its own sample only (bitwise) — what makes
    sample-sharding across GPUs exact"""
    m, sd = make_cfg(8)
    x, rf, mk = _cond_inputs(5, 32, 200)
    t = torch.full((5,), 499)
    c = torch.zeros(5, dtype=torch.long)
    full = m(g(x), g(t), g(c), g(rf), g(mk), cond_drop_prob=0.)
    part = m(g(x[3:]), g(t[3:]), g(c[3:]), g(rf[3:]), g(mk[3:]), cond_drop_prob=0.)
    assert torch.equal(full[3:], part)
    again = m(g(x), g(t), g(c), g(rf), g(mk), cond_drop_prob=0.)
    assert torch.equal(full, again)                       # deterministic: no atomics anywhere


def test_fullsize_rows_independent_and_cfg_modes_agree():
    """dim 64 at 128x128 (the bench geometry; fused attention, fp16-piece convolutions with per-tile block scales):
    a sample's output is bitwise the same alone, inside a larger batch, and whichever way the two CFG passes are
    scheduled ('batched' = one 2B-row launch sequence, 'streams' = two B-row sequences on two HIP streams)"""
    m, sd = make_cfg(64)
    x, rf, mk = _cond_inputs(3, 128, 500)
    t = torch.full((3,), 967)
    c = torch.zeros(3, dtype=torch.long)
    full = m(g(x), g(t), g(c), g(rf), g(mk), cond_drop_prob=0.)
    part = m(g(x[2:]), g(t[2:]), g(c[2:]), g(rf[2:]), g(mk[2:]), cond_drop_prob=0.)
    assert torch.equal(full[2:], part)
    outs = {}
    for mode in ('batched', 'streams'):
        m.cfg_mode = mode
        torch.manual_seed(5)          # the conditional pass draws its class-dropout mask (CFG:404 -> CFG:422), like the reference
        outs[mode] = m.forward_with_cond_scale(g(x), g(t), g(c), g(rf), g(mk), cond_scale=3.).clone()
    m.cfg_mode, m.dedup_dropped_rows = 'batched', True     # opt-in: dropped rows of the conditional pass computed once
    torch.manual_seed(5)
    outs['dedup'] = m.forward_with_cond_scale(g(x), g(t), g(c), g(rf), g(mk), cond_scale=3.).clone()
    m.dedup_dropped_rows = False
    assert torch.equal(outs['batched'], outs['dedup'])
    assert torch.isfinite(outs['batched']).all()
    assert torch.equal(outs['batched'], outs['streams'])


@pytest.mark.parametrize('dim,size', [(64, 128), (8, 16), (16, 40)])
def test_final_conv_fused_into_last_res_conv_is_bitwise(dim, size, monkeypatch):
    """final_conv (CFG:341, 471-472) rides on the last ResnetBlock's res_conv launch (DmhConv.fin_*): the forward is bit for
    bit the one with final_conv as its own launch, and a tapped run (which stores the block output) agrees with both"""
    from dmhomo_amd import engine as E
    m, sd = make_cfg(dim)
    x, rf, mk = _cond_inputs(2, size, 510)
    t, c = torch.tensor([900, 17]), torch.zeros(2, dtype=torch.long)
    assert E.FUSED_FINAL
    fused = m(g(x), g(t), g(c), g(rf), g(mk), cond_drop_prob=0.).clone()
    monkeypatch.setattr(E, 'FUSED_FINAL', False)
    plain = m(g(x), g(t), g(c), g(rf), g(mk), cond_drop_prob=0.).clone()
    assert torch.equal(fused, plain)
    monkeypatch.setattr(E, 'FUSED_FINAL', True)
    taps = {}
    tapped = m._run(g(x), g(t), g(c), g(rf), g(mk), [None], taps=taps)
    assert torch.equal(tapped, fused) and taps['final_res_block'] is not None


def test_unet_weight_update_is_picked_up():
    m, sd = make_cfg(8)
    x, rf, mk = _cond_inputs(1, 16, 300)
    t, c = torch.tensor([5]), torch.zeros(1, dtype=torch.long)
    a = m(g(x), g(t), g(c), g(rf), g(mk), cond_drop_prob=0.).clone()
    with torch.no_grad():
        m.final_conv.bias.add_(1.0)
    b = m(g(x), g(t), g(c), g(rf), g(mk), cond_drop_prob=0.)
    torch.testing.assert_close(b, a + 1.0, rtol=0, atol=1e-5)


def test_cpu_tensor_fails_loudly():
    m, sd = make_cfg(8)
    x, rf, mk = _cond_inputs(1, 16, 300)
    with pytest.raises(RuntimeError):
        m(x, torch.tensor([5]), torch.zeros(1, dtype=torch.long), rf, mk)


# ------------------------------------------------------------------------------- samplers
@pytest.mark.parametrize('obj', ['pred_x0', 'pred_noise', 'pred_v'])
def test_ddim_trace_vs_golden(golden_dir, obj):
    """F5: the reference's own sample() run (S=4, 16x16) replaying its recorded RNG draws"""
    from dmhomo_amd import cfg
    gd = load(golden_dir, 'ddim_trace')
    m, sd = make_cfg(8)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective=obj).to(dev())
    d.rng = ReplayDeviceRng([gd[f'{obj}.draw{i}'] for i in range(8)])
    trace = []
    shape = (2, 6, 16, 16)
    from dmhomo_amd import ops
    rgbn = ops.affine(g(T(gd['rgb_flow01'])), 2., -1.)
    img, mk, fl = d._ddim_sample(g(T(gd['classes'])), rgbn, g(T(gd['flow'])), g(T(gd['mask'])), shape, trace=trace)
    tol = 3e-4                                        # measured: x_start <= 3.9e-5, img <= 1.4e-5 (all three objectives)
    for i, st in enumerate(trace):
        close(f'{obj} x_start{i}', st['x_start'].cpu(), T(gd[f'{obj}.x_start{i}']), rtol=0, atol=tol)
    close(f'{obj} img', img.cpu(), T(gd[f'{obj}.img']), rtol=0, atol=tol)
    assert d.rng.i == 8
    # sample() = rgb_flow*2-1 + ddim_sample, same result
    d.rng = ReplayDeviceRng([gd[f'{obj}.draw{i}'] for i in range(8)])
    img2, mk2, fl2 = d.sample(g(T(gd['classes'])), g(T(gd['rgb_flow01'])), g(T(gd['flow'])), g(T(gd['mask'])))
    assert torch.equal(img2, img) and torch.equal(mk2.cpu(), T(gd['mask'])) and torch.equal(fl2.cpu(), T(gd['flow']))
    assert float(img.min()) >= 0 and float(img.max()) <= 1


@pytest.mark.parametrize('tag', ['nosc', 'sc'])
def test_ddpm_trace_vs_golden(golden_dir, tag):
    """F6: DDP p_sample_loop (T=10), ddim_sample (S=4) and a single p_sample"""
    from dmhomo_amd import ddpm
    gd = load(golden_dir, 'ddpm_trace')
    m, sd = make_ddp(8, tag == 'sc')
    d = ddpm.GaussianDiffusion(m, image_size=16, timesteps=10, objective='pred_noise').to(dev())
    d.rng = ReplayDeviceRng([gd[f'{tag}.ddpm.draw{i}'] for i in range(10)])
    # 10 ancestral steps with pred_noise (x_start = a*x - b*eps, gains up to ~20 at the first steps): measured 1.1e-5
    # (nosc) / 1.2e-4 (sc: the self-conditioning input feeds each step's error back into the network)
    close(f'ddpm {tag}', d.sample(batch_size=2).cpu(), T(gd[f'{tag}.ddpm.img']), rtol=0, atol=1e-3 if tag == 'sc' else 1e-4)
    d2 = ddpm.GaussianDiffusion(m, image_size=16, timesteps=10, sampling_timesteps=4, objective='pred_x0').to(dev())
    d2.rng = ReplayDeviceRng([gd[f'{tag}.ddim.draw{i}'] for i in range(4)])
    got, want = d2.sample(batch_size=2).cpu(), T(gd[f'{tag}.ddim.img'])
    close(f'ddp ddim {tag} image channel', got[:, :-2], want[:, :-2], rtol=0, atol=2e-4)       # [0, 1]
    close(f'ddp ddim {tag} flow channels (x512)', got[:, -2:], want[:, -2:], rtol=0, atol=6e-2)  # measured 6.7e-3 of +-512
    if tag == 'nosc':
        d.rng = ReplayDeviceRng([gd['p_sample.noise']])
        img, xs = d.p_sample(g(T(gd['p_sample.x'])), 5)
        close('p_sample img', img.cpu(), T(gd['p_sample.img']), rtol=0, atol=5e-5)
        close('p_sample x_start', xs.cpu(), T(gd['p_sample.x_start']), rtol=0, atol=5e-5)


def test_sample_fullsize_vs_oracle_and_properties():
    """config-1 geometry (dim=64, 128x128, bs=2, s_step=4): HIP sample() vs the oracle on replayed noise.
    timesteps=100 keeps the first DDIM jump (99 -> 74) out of the regime where the reference's
    c = sqrt(1 - a' - sigma^2) is pure fp32 cancellation noise (999 -> 749 at T=1000: 0 or NaN by host)."""
    from dmhomo_amd import cfg
    from dmhomo_amd import ops
    m, sd = make_cfg(64)
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=100, sampling_timesteps=4, objective='pred_x0').to(dev())
    B = 2
    _, rf, mk = _cond_inputs(B, 128, 400)
    rf01 = (rf + 1) / 2
    flow = rand((B, 2, 128, 128), 403)
    c = torch.zeros(B, dtype=torch.long)
    torch.manual_seed(99)
    rec = OD.RecordRng()
    buf = OD.schedule_buffers(100, 'cosine')
    with torch.no_grad():
        ref, _, _ = OD.cfg_sample(sd, buf, c, rf01, flow, mk, image_size=128, channels=6, sampling_timesteps=4,
                                  objective='pred_x0', rng=rec)
    d.rng = ReplayDeviceRng(rec.draws)
    img, mk2, fl2 = d.sample(g(c), g(rf01), g(flow), g(mk))
    close('sample full', img.cpu(), ref, rtol=0, atol=4e-4)          # measured 3.7e-5
    u8 = ops.to_uint8(img).cpu().numpy().astype(np.int32)
    ru8 = (ref.numpy() * 255).astype(np.uint8).astype(np.int32)
    assert np.abs(u8 - ru8).max() <= 1
    assert float(img.min()) >= 0 and float(img.max()) <= 1


def test_sample_config0_as_written():
    """BASELINE configs[0] literally: dim=64, 128x128, bs=2, s_step=4 at the reference's timesteps=1000 (CFG:669-711).  The
    first DDIM jump (999 -> 749) has alpha ~ 2.4e-9, so the reference's radicand 1 - alpha' - sigma^2 (CFG:701) is fp32
    cancellation noise: 0 or -6e-8 (-> c = NaN and an all-NaN image) depending on how the host's libm rounds; the product
    clamps it at 0 (cfg.py _ddim_coef, the documented deviation).  Checked here: the oracle's own radicand on THIS host is
    printed; the product is finite, in [0, 1], graph == eager bitwise, draw count as the reference's; and wherever the
    oracle is finite (everywhere or nowhere) the image agrees with it within the sampler gate and +-1 uint8 LSB.  If the
    oracle is NaN on this host the comparison runs against the oracle with the same clamp (the only difference)."""
    from dmhomo_amd import cfg
    from dmhomo_amd import ops
    m, sd = make_cfg(64)
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=4, objective='pred_x0').to(dev())
    B = 2
    _, rf, mk = _cond_inputs(B, 128, 410)
    rf01, flow, c = (rf + 1) / 2, rand((B, 2, 128, 128), 413), torch.zeros(B, dtype=torch.long)
    buf = OD.schedule_buffers(1000, 'cosine')
    ac = buf['alphas_cumprod']
    a, an = ac[999], ac[749]
    sigma = ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
    radicand = float(1 - an - sigma ** 2)
    print(f'[parity] configs[0]: the oracle\'s first-jump radicand on this host = {radicand!r} '
          f'(alpha = {float(a):.3e}, alpha_next = {float(an):.6f}, sigma = {float(sigma):.6f})')
    torch.manual_seed(99)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rec = OD.RecordRng()
    with torch.no_grad():
        ref, _, _ = OD.cfg_sample(sd, buf, c, rf01, flow, mk, image_size=128, channels=6, sampling_timesteps=4,
                                  objective='pred_x0', rng=rec)
    assert len(rec.draws) == 1 + 4 + 3                      # initial noise, 4 class-dropout draws, 3 step noises
    finite = torch.isfinite(ref)
    assert bool(finite.all()) or not bool(finite.any())     # NaN in c poisons every element of the first update
    if not bool(finite.all()):
        assert radicand < 0
        orig = OD._ddim_update

        def clamped(buf_, x_start, pred_noise, time, time_next, eta, noise):   # the reference's update with the radicand at 0
            alpha, alpha_next = buf_['alphas_cumprod'][time], buf_['alphas_cumprod'][time_next]
            sg = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            cc = (1 - alpha_next - sg ** 2).clamp(min=0).sqrt()
            return x_start * alpha_next.sqrt() + cc * pred_noise + sg * noise
        OD._ddim_update = clamped
        try:
            with torch.no_grad():
                ref, _, _ = OD.cfg_sample(sd, buf, c, rf01, flow, mk, image_size=128, channels=6, sampling_timesteps=4,
                                          objective='pred_x0', rng=OD.ReplayRng(rec.draws))
        finally:
            OD._ddim_update = orig
        print('[parity] configs[0]: the oracle is NaN on this host; compared with the oracle under the radicand clamp')
    d.rng = ReplayDeviceRng(rec.draws)
    img, mk2, fl2 = d.sample(g(c), g(rf01), g(flow), g(mk))
    assert d.rng.i == len(rec.draws)
    assert bool(torch.isfinite(img).all()) and float(img.min()) >= 0 and float(img.max()) <= 1
    assert torch.equal(mk2.cpu(), mk) and torch.equal(fl2.cpu(), flow)
    close('configs[0] sample (T=1000, S=4)', img.cpu(), ref, rtol=0, atol=4e-4)
    u8 = ops.to_uint8(img).cpu().numpy().astype(np.int32)
    ru8 = (ref.numpy() * 255).astype(np.uint8).astype(np.int32)
    assert np.abs(u8 - ru8).max() <= 1
    # the captured step against the eager loop, on the device generator (bitwise)
    d.rng = cfg.DeviceRng().key_by_sample(7, range(B), dev())
    eager = d.sample(g(c), g(rf01), g(flow), g(mk))[0]
    d.hip_graph = True
    d.rng.key_by_sample(7, range(B), dev())
    assert torch.equal(d.sample(g(c), g(rf01), g(flow), g(mk))[0], eager)
    d.hip_graph = False


@pytest.mark.parametrize('S', [16, 32])
def test_ddim_trace_s32_vs_golden(golden_dir, S):
    """F5 at the README's s_step = 32, T = 1000 (tests/golden/make_golden_r2.py): the reference's own 32-step sample()
    run replayed draw for draw (64 draws), per-step x_start and the final image"""
    from dmhomo_amd import cfg
    gd = load(golden_dir, 'ddim_trace_s32')
    m, sd = make_cfg(8)
    d = cfg.GaussianDiffusion(m, image_size=S, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    d.rng = ReplayDeviceRng([gd[f's{S}.draw{i}'] for i in range(64)])
    trace = []
    from dmhomo_amd import ops
    rgbn = ops.affine(g(T(gd[f's{S}.rgb_flow01'])), 2., -1.)
    img, _, _ = d._ddim_sample(g(T(gd[f's{S}.classes'])), rgbn, g(T(gd[f's{S}.flow'])), g(T(gd[f's{S}.mask'])),
                              (2, 6, S, S), trace=trace)
    assert d.rng.i == 64 and len(trace) == 32
    drift = [float((st['x_start'].cpu() - T(gd[f's{S}.x_start{i}'])).abs().max()) for i, st in enumerate(trace)]
    print(f'[parity] ddim S=32 {S}x{S}: per-step max|x_start - reference| = ' + ' '.join(f'{e:.1e}' for e in drift))
    assert max(drift) <= 4e-4, max(drift)
    close(f'ddim S=32 {S}x{S} img', img.cpu(), T(gd[f's{S}.img']), rtol=0, atol=4e-4)


@pytest.mark.parametrize('kw', [dict(eta=0.0), dict(eta=0.5), dict(schedule='linear'), dict(T=200, S=7), dict(cond_scale=1.0),
                                dict(cond_scale=0.5), dict(cond_scale=7.0), dict(drop=0.0), dict(drop=1.0), dict(drop=0.2),
                                dict(objective='pred_v', eta=0.3, schedule='linear'), dict(S=1), dict(T=50, S=49)],
                         ids=lambda kw: ','.join(f'{k}={v}' for k, v in kw.items()))
def test_sampler_argument_variants_vs_oracle(kw):
    """GaussianDiffusion / Unet arguments away from the DGM's values — ddim_sampling_eta 0 / 0.5, the linear schedule, other
    (T, S) incl. S = 1 and S = T - 1, cond_scale 1 (no guidance: one pass, one draw fewer per step) / 0.5 / 7, cond_drop_prob
    0 / 1 (no draw) / 0.2, pred_v — the sampler against the oracle on replayed draws (the number of draws consumed must
    match too), and the per-step graph against the eager loop, bitwise.  T = 100 by default: at T = 1000 a 5-step schedule's
    first jump is fp32 cancellation noise in the reference itself (deviations)."""
    from dmhomo_amd import cfg
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    T, S = kw.get('T', 100), kw.get('S', 5)
    m = cfg.Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1, cond_drop_prob=kw.get('drop', 0.5))
    sd = det_state_dict(shapes_of(m), 0)
    m.load_state_dict(sd)
    m = m.to(dev())
    obj, sched, eta, cs = kw.get('objective', 'pred_x0'), kw.get('schedule', 'cosine'), kw.get('eta', 1.), kw.get('cond_scale', 3.)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=T, sampling_timesteps=S, objective=obj, beta_schedule=sched,
                              ddim_sampling_eta=eta).to(dev())
    B = 2
    gen = torch.Generator().manual_seed(9)
    rf01 = torch.rand(B, 3, 16, 16, generator=gen)
    mk = (torch.rand(B, 1, 16, 16, generator=gen) > 0.4).float()
    fl = torch.randn(B, 2, 16, 16, generator=gen)
    c = torch.zeros(B, dtype=torch.long)
    torch.manual_seed(4)
    rec = OD.RecordRng()
    with torch.no_grad():
        ref, _, _ = OD.cfg_sample(sd, OD.schedule_buffers(T, sched), c, rf01, fl, mk, image_size=16, channels=6,
                                  sampling_timesteps=S, objective=obj, cond_scale=cs, cond_drop_prob=kw.get('drop', 0.5), eta=eta,
                                  rng=rec)
    assert torch.isfinite(ref).all()
    d.rng = ReplayDeviceRng(rec.draws)
    img, _, _ = d.sample(g(c), g(rf01), g(fl), g(mk), cond_scale=cs)
    assert d.rng.i == len(rec.draws)                      # the same number of draws, in the same order
    close(f'sampler {kw}', img.cpu(), ref, rtol=0, atol=4e-4)
    from dmhomo_amd.cfg import DeviceRng
    d.rng = DeviceRng()
    outs = []
    for graph in (False, True):
        d.hip_graph = graph
        torch.manual_seed(11)
        outs.append(d.sample(g(c), g(rf01), g(fl), g(mk), cond_scale=cs)[0].clone())
    assert torch.equal(outs[0], outs[1])


def _fullsize_s32_inputs(B):
    _, rf, mk = _cond_inputs(B, 128, 700)
    return (rf + 1) / 2, rand((B, 2, 128, 128), 703), mk, torch.zeros(B, dtype=torch.long)


def test_sample_fullsize_s32_vs_oracle():
    """BASELINE configs[1] at its real depth: dim 64, 128x128, timesteps = 1000, sampling_timesteps = 32 (the hot loop
    CFG:683-707 as dgm_sample.py:34 drives it), B = 2, against the oracle on replayed noise.  Error growth over 32
    full-width denoise steps is what this pins (SURVEY §7 'hard parts'); the per-step drift is printed."""
    from dmhomo_amd import cfg, ops
    m, sd = make_cfg(64)
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    B = 2
    rf01, flow, mk, c = _fullsize_s32_inputs(B)
    torch.manual_seed(99)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rec = OD.RecordRng()
    rtrace = []
    with torch.no_grad():
        ref, _, _ = OD.cfg_sample(sd, OD.schedule_buffers(1000, 'cosine'), c, rf01, flow, mk, image_size=128, channels=6,
                                  sampling_timesteps=32, objective='pred_x0', rng=rec, trace=rtrace)
    assert len(rec.draws) == 64
    d.rng = ReplayDeviceRng(rec.draws)
    trace = []
    rgbn = ops.affine(g(rf01), 2., -1.)
    img, _, _ = d._ddim_sample(g(c), rgbn, g(flow), g(mk), (B, 6, 128, 128), trace=trace)
    drift = [float((a['x_start'].cpu() - b['x_start']).abs().max()) for a, b in zip(trace, rtrace)]
    print('[parity] sample S=32 full size: per-step max|x_start - oracle| = ' + ' '.join(f'{e:.1e}' for e in drift))
    close('sample S=32 full', img.cpu(), ref, rtol=0, atol=2e-4)
    u8 = ops.to_uint8(img).cpu().numpy().astype(np.int32)
    ru8 = (ref.numpy() * 255).astype(np.uint8).astype(np.int32)
    assert np.abs(u8 - ru8).max() <= 1
    assert float(img.min()) >= 0 and float(img.max()) <= 1


def test_sample_bs25_s32_rows_match_bs2():
    """configs[1] at its real batch and depth (bs = 25, s_step = 32, 128x128, the bench's 'streams' CFG mode) with noise
    keyed by sample index: rows 0-1 are BITWISE the rows of a bs = 2 run (what sample-sharding across GPUs relies on),
    everything is finite and inside [0, 1]."""
    from dmhomo_amd import cfg
    from dmhomo_amd.distributed import SampleIndexedRng
    m, sd = make_cfg(64)
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    rf01, flow, mk, c = (g(v) for v in _fullsize_s32_inputs(25))
    outs = {}
    for B, mode in ((25, 'streams'), (2, 'batched')):
        m.cfg_mode = mode
        d.rng = SampleIndexedRng(7, range(B), dev())
        outs[B], _, _ = d.sample(c[:B], rf01[:B].contiguous(), flow[:B].contiguous(), mk[:B].contiguous())
    m.cfg_mode = 'batched'
    assert outs[25].shape == (25, 6, 128, 128) and torch.isfinite(outs[25]).all()
    assert float(outs[25].min()) >= 0 and float(outs[25].max()) <= 1
    assert torch.equal(outs[25][:2], outs[2])


def test_sample_bs25_s32_high_rows_vs_oracle():
    """configs[1] at its real batch (bs = 25, s_step = 32, 128x128, 'streams'): rows 13 and 24 of the 25 against the
    ORACLE run on those two samples alone with the same draws (the draws of the 25-row run are recorded and the two rows'
    slices replayed) — an oracle number for rows beyond the first two of the batch (round 2 checked rows 0-1 against the
    oracle and the rest HIP against HIP)"""
    from dmhomo_amd import cfg
    from dmhomo_amd.distributed import SampleIndexedRng
    m, sd = make_cfg(64)
    m.cfg_mode = 'streams'
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    rf01, flow, mk, c = _fullsize_s32_inputs(25)
    rows = [13, 24]

    class Rec:
        def __init__(self, inner):
            self.inner, self.draws = inner, []

        def randn(self, shape, device):
            v = self.inner.randn(shape, device)
            self.draws.append(v[rows].cpu())
            return v

        def uniform(self, n, device):
            v = self.inner.uniform(n, device)
            self.draws.append(v[rows].cpu())
            return v
    d.rng = rec = Rec(SampleIndexedRng(11, range(25), dev()))
    img, _, _ = d.sample(g(c), g(rf01), g(flow), g(mk))
    m.cfg_mode = 'batched'
    assert len(rec.draws) == 1 + 32 + 31                      # initial noise, 32 class-dropout draws, 31 step noises
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref, _, _ = OD.cfg_sample(sd, OD.schedule_buffers(1000, 'cosine'), c[rows], rf01[rows], flow[rows], mk[rows],
                                  image_size=128, channels=6, sampling_timesteps=32, objective='pred_x0',
                                  rng=OD.ReplayRng(rec.draws))
    close('sample bs=25 S=32 rows 13, 24 vs oracle', img[rows].cpu(), ref, rtol=0, atol=2e-4)


@pytest.mark.parametrize('mode,S,size', [('batched', 6, 32), ('streams', 6, 32), ('streams', 250, 16), ('batched', 1, 16)])
def test_sample_hip_graph_equals_eager(mode, S, size):
    """GaussianDiffusion.hip_graph: ONE denoise step of the sampling loop (the hot loop CFG:683-707)
    captured into a HIP graph whose coefficients / timestep come from device tables, replayed S times (+ the last step's
    own graph): bitwise what the eager launch sequence computes from the same device RNG state — on the capturing call
    itself (the warm-up must not consume random numbers), when the inputs change between replays, at s_step = 250
    (BASELINE configs[4]'s depth), and again after a weight update (re-capture on the new weight version) or a change of
    ddim_sampling_eta (baked into the step table)."""
    from dmhomo_amd import cfg
    m, sd = make_cfg(8)
    m.cfg_mode = mode
    d = cfg.GaussianDiffusion(m, image_size=size, timesteps=1000, sampling_timesteps=S, objective='pred_x0').to(dev())
    _, rf, mk = _cond_inputs(3, size, 900)
    rf01, flow, c = g((rf + 1) / 2), g(rand((3, 2, size, size), 903)), g(torch.zeros(3, dtype=torch.long))
    mk = g(mk)

    def run(graph, seed, rf_in):
        d.hip_graph = graph
        torch.manual_seed(seed)
        img, mo, fo = d.sample(c, rf_in, flow, mk)
        tail = torch.rand(4, device=dev())               # the generator is left where the eager path leaves it
        return torch.cat([img.flatten(), tail])
    eager5, eager6 = run(False, 5, rf01), run(False, 6, 1 - rf01)
    assert torch.equal(run(True, 5, rf01), eager5)       # the capturing call itself
    g0 = d.__dict__['_graph_state']['graph']
    assert torch.equal(run(True, 6, 1 - rf01), eager6)   # new inputs, new RNG state, same graph
    assert torch.equal(run(True, 5, rf01), eager5)
    assert d.__dict__['_graph_state']['graph'] is g0
    d.ddim_sampling_eta = 0.5                            # enters the step table: a stale replay would keep eta = 1
    e_eta = run(False, 5, rf01)
    assert not torch.equal(e_eta, eager5) or S == 1
    assert torch.equal(run(True, 5, rf01), e_eta)
    d.ddim_sampling_eta = 1.
    with torch.no_grad():
        m.final_conv.bias.add_(0.25)                     # a new weight version: re-capture, not a stale replay
    e = run(False, 5, rf01)
    assert not torch.equal(e, eager5)
    g_old = d.__dict__['_graph_state']['graph']
    assert torch.equal(run(True, 5, rf01), e)
    assert d.__dict__['_graph_state']['graph'] is not g_old
    d.hip_graph = False


def test_graph_cache_keeps_alternating_batch_shapes():
    """The captured steps live in a small LRU (GaussianDiffusion.graph_cache_size = 4): a job that alternates a full and a
    short batch — the last batch of every epoch of scripts/dgm_sample.py's loader (DDP:1746-1752 keeps it) — captures each
    shape ONCE (bs 5 / bs 3 / bs 5 / bs 3: two captures, the second round replays), every call bitwise the eager call, with
    the keyed generator a short batch reads through a view (cfg.DeviceRng.ids_for); the cache evicts least-recently-used."""
    from dmhomo_amd import cfg
    m, sd = make_cfg(8)
    m.cfg_mode, m.dedup_dropped_rows = 'streams', True
    size = 16
    d = cfg.GaussianDiffusion(m, image_size=size, timesteps=1000, sampling_timesteps=5, objective='pred_x0').to(dev())
    _, rf, mk = _cond_inputs(5, size, 910)
    rf01, flow, c, mk = g((rf + 1) / 2), g(rand((5, 2, size, size), 913)), g(torch.zeros(5, dtype=torch.long)), g(mk)

    def run(graph, n, seed):
        d.hip_graph = graph
        d.rng.key_by_sample(seed, range(100, 105), dev())
        return d.sample(c[:n], rf01[:n], flow[:n], mk[:n])[0]
    d.rng = cfg.DeviceRng()
    eager = {(n, s): run(False, n, s) for n in (5, 3, 2) for s in (1, 2)}
    assert d.graph_captures == 0
    for rnd, s in ((0, 1), (1, 2)):
        for n in (5, 3):
            assert torch.equal(run(True, n, s), eager[(n, s)]), (rnd, n)
        assert d.graph_captures == 2, d.graph_captures          # the second round replays both
    st5 = d.__dict__['_graph_states']
    assert len(st5) == 2
    d.graph_cache_size = 2                                      # a third shape evicts the least recently used (bs 5)
    assert torch.equal(run(True, 2, 1), eager[(2, 1)]) and d.graph_captures == 3 and len(st5) == 2
    assert torch.equal(run(True, 3, 1), eager[(3, 1)]) and d.graph_captures == 3      # bs 3 survived
    assert torch.equal(run(True, 5, 2), eager[(5, 2)]) and d.graph_captures == 4      # bs 5 did not
    d.hip_graph = False


def test_exact_fp32_conv_variant_in_child_process():
    """DMH_CONV3_VARIANT=6 — the exact-fp32 Winograd 3x3 path the bench line advertises as the alternative to the
    fp16-piece kernel — is read once per process, so it is exercised in a fresh child: conv2d kernel parity and the
    full-size UNet vs the oracle under it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMH_CONV3_VARIANT='6')
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        os.path.join(root, 'tests', 'test_gpu_kernels.py') + '::test_conv2d',
                        os.path.join(root, 'tests', 'test_gpu_unet.py') + '::test_unet_cfg_fullsize_vs_oracle'],
                       capture_output=True, text=True, env=env, cwd=root, timeout=1500)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_state_dict_roundtrip_through_trainer(tmp_path):
    """checkpoint dict layout of DDP:1786-1802 + Trainer.sample record format (G6)"""
    from dmhomo_amd import cfg, ddpm
    m, sd = make_cfg(8)
    d = cfg.GaussianDiffusion(m, image_size=32, timesteps=1000, sampling_timesteps=2, objective='pred_x0').to(dev())
    tr = ddpm.Trainer(d, 'DGM_Conditions', train_batch_size=3, results_folder=str(tmp_path), num_samples=4)
    ret = tr.sample(0, 0, step=1)
    assert ret['imgs'].dtype == np.uint8 and ret['imgs'].shape == (3, 6, 32, 32)
    assert ret['homos'].dtype == np.float64 and ret['homos'].shape == (3, 3, 3)
    tr.save(7)
    ck = torch.load(str(tmp_path / 'model-7.pt'), map_location='cpu')
    assert set(ck) == {'step', 'model', 'opt', 'ema', 'scaler', 'version'}
    m2, _ = make_cfg(8, seed=5)
    d2 = cfg.GaussianDiffusion(m2, image_size=32, timesteps=1000, sampling_timesteps=2, objective='pred_x0').to(dev())
    tr2 = ddpm.Trainer(d2, 'DGM_Conditions', train_batch_size=3, results_folder=str(tmp_path))
    tr2.load(7)
    for (k, a), (_, b) in zip(d.state_dict().items(), d2.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize('obj,lt', [('pred_x0', 'l1'), ('pred_noise', 'l2'), ('pred_v', 'l1')])
def test_p_losses_forward_vs_golden(golden_dir, obj, lt):
    """F8: the reference's own p_losses value (CFG:770-806), forward only"""
    from dmhomo_amd import cfg
    gd = load(golden_dir, 'train_forward')
    m, sd = make_cfg(8)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective=obj,
                              loss_type=lt).to(dev())
    img12 = T(gd['img12'])
    data, mk, rf, fl = img12[:, :6] * 2 - 1, img12[:, 6:7], img12[:, -5:-2] * 2 - 1, img12[:, -2:]
    m.rng = ReplayDeviceRng([torch.where(T(gd[f'{obj}.{lt}.keep']), 0.25, 0.75)])
    loss = d.p_losses(g(data), g(T(gd['t'])), classes=g(T(gd['classes'])), rgb_flow=g(rf.contiguous()),
                      flow=g(fl.contiguous()), mask=g(mk.contiguous()), noise=g(T(gd['noise'])))
    want = float(gd[f'{obj}.{lt}.loss'])
    print(f'[parity] p_losses {obj}/{lt}: got {float(loss):.7f} want {want:.7f}')
    assert abs(float(loss) - want) <= 2e-4 * max(1.0, abs(want))
    # GaussianDiffusion.forward: 12-channel split + random t (device RNG) -> finite scalar
    m.rng = cfg.DeviceRng()
    val = d(g(img12), classes=g(T(gd['classes'])))
    assert val.ndim == 0 and bool(torch.isfinite(val))


def test_unet_stress_geometry_vs_oracle():
    """BASELINE config 5 geometry: dim=128, 256x256 (1024-key bottleneck attention, 65536-pixel linear attention)"""
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=128, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    sd = det_state_dict(shapes_of(m), 3)
    m.load_state_dict(sd)
    m = m.to(dev())
    x, rf, mk = _cond_inputs(1, 256, 500)
    t, c = torch.tensor([499]), torch.zeros(1, dtype=torch.long)
    rtaps, taps = {}, {}
    with torch.no_grad():
        ref = OU.cfg_unet_forward(sd, x, t, c, rf, mk, None, taps=rtaps)
    out = m._run(g(x), g(t), g(c), g(rf), g(mk), [None], taps=taps)
    worst = max(report('stress ' + k, nchw(taps[k]), rtaps[k])[1] for k in ('downs.0.2', 'mid_attn', 'ups.3.3'))
    assert worst < LAYER_REL, worst
    close_rel('stress out', out.cpu(), ref, OUT_REL)


def test_unet_readme_geometry_256_vs_oracle():
    """the README's inference geometry (DGM/dgm_sample.py:28-38 hard-codes image_size = 256 with dim = 64): 65 536-pixel
    LinearAttention at the first level, 32x32 = 1024-key bottleneck attention; B = 2 (one kept, one dropped class row),
    per-layer taps and the output against the oracle"""
    m, sd = make_cfg(64)
    x, rf, mk = _cond_inputs(2, 256, 800)
    t = torch.tensor([967, 30])
    c = torch.zeros(2, dtype=torch.long)
    keep = torch.tensor([True, False])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rtaps, taps = {}, {}
    with torch.no_grad():
        ref = OU.cfg_unet_forward(sd, x, t, c, rf, mk, keep, taps=rtaps)
    out = m._run(g(x), g(t), g(c), g(rf), g(mk), [g(keep.to(torch.uint8))], taps=taps)
    worst = max(report('256/dim64 ' + k, nchw(taps[k]), rtaps[k])[1] for k in rtaps)
    assert worst < LAYER_REL, worst
    close_rel('256/dim64 out', out.cpu(), ref, OUT_REL)


@pytest.mark.parametrize('dim,tag', [(64, 'README geometry 256x256'), (128, 'BASELINE configs[4] geometry')])
def test_sample_256_vs_oracle(dim, tag):
    """the sampler (not just one forward) at 256x256: dim 64 (README / dgm_sample.py) and dim 128 (configs[4]), B = 1,
    4 DDIM steps of a T = 100 schedule (see test_sample_fullsize_vs_oracle for why not 1000 -> 4) against the oracle on
    replayed noise; uint8 record within 1 LSB"""
    from dmhomo_amd import cfg, ops
    m = cfg.Unet(dim=dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    sd = det_state_dict(shapes_of(m), 3)
    m.load_state_dict(sd)
    m = m.to(dev())
    d = cfg.GaussianDiffusion(m, image_size=256, timesteps=100, sampling_timesteps=4, objective='pred_x0').to(dev())
    B = 1
    _, rf, mk = _cond_inputs(B, 256, 820)
    rf01, flow, c = (rf + 1) / 2, rand((B, 2, 256, 256), 823), torch.zeros(B, dtype=torch.long)
    torch.manual_seed(99)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rec = OD.RecordRng()
    with torch.no_grad():
        ref, _, _ = OD.cfg_sample(sd, OD.schedule_buffers(100, 'cosine'), c, rf01, flow, mk, image_size=256, channels=6,
                                  sampling_timesteps=4, objective='pred_x0', rng=rec)
    d.rng = ReplayDeviceRng(rec.draws)
    img, _, _ = d.sample(g(c), g(rf01), g(flow), g(mk))
    close(f'sample 256 dim {dim} ({tag})', img.cpu(), ref, rtol=0, atol=4e-4)
    u8 = ops.to_uint8(img).cpu().numpy().astype(np.int32)
    ru8 = (ref.numpy() * 255).astype(np.uint8).astype(np.int32)
    assert np.abs(u8 - ru8).max() <= 1


def test_stress_config_s250_graph_equals_eager():
    """BASELINE configs[4] as a sampler run: dim 128, 256x256, s_step = 250 of T = 1000, 'streams' CFG mode, B = 1 —
    250 replays of the captured denoise step give bitwise the eager loop's images; finite and inside [0, 1]"""
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=128, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    m.load_state_dict(det_state_dict(shapes_of(m), 3))
    m = m.to(dev())
    m.cfg_mode = 'streams'
    d = cfg.GaussianDiffusion(m, image_size=256, timesteps=1000, sampling_timesteps=250, objective='pred_x0').to(dev())
    _, rf, mk = _cond_inputs(1, 256, 840)
    rf01, flow, c = g((rf + 1) / 2), g(rand((1, 2, 256, 256), 843)), g(torch.zeros(1, dtype=torch.long))
    outs = []
    for graph in (False, True):
        d.hip_graph = graph
        torch.manual_seed(17)
        outs.append(d.sample(c, rf01, flow, g(mk))[0].clone())
    d.hip_graph = False
    assert torch.isfinite(outs[0]).all() and float(outs[0].min()) >= 0 and float(outs[0].max()) <= 1
    assert torch.equal(outs[0], outs[1])


def test_stress_config_at_its_stated_batch():
    """BASELINE configs[4] AS STATED: dim 128, 256x256, s_step = 250, bs = 8, 'streams' CFG mode, per-step HIP graph, noise
    keyed by sample index — the full 250-step run at B = 8; rows 0 and 7 are BITWISE the rows of B = 1 runs of those two
    samples (250 replays of a 2-row step each), everything finite and inside [0, 1], rows distinct."""
    import time
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=128, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    m.load_state_dict(det_state_dict(shapes_of(m), 3))
    m = m.to(dev())
    m.cfg_mode = 'streams'
    d = cfg.GaussianDiffusion(m, image_size=256, timesteps=1000, sampling_timesteps=250, objective='pred_x0').to(dev())
    d.hip_graph = True
    B = 8
    _, rf, mk = _cond_inputs(B, 256, 850)
    rf01, flow, c, mk = g((rf + 1) / 2), g(rand((B, 2, 256, 256), 853)), g(torch.zeros(B, dtype=torch.long)), g(mk)

    def run(rows):
        d.rng.key_by_sample(23, rows, dev())
        sel = torch.tensor(list(rows), device=dev())
        t0 = time.perf_counter()
        img = d.sample(c[sel], rf01[sel].contiguous(), flow[sel].contiguous(), mk[sel].contiguous())[0].clone()
        torch.cuda.synchronize()
        return img, time.perf_counter() - t0
    whole, sec = run(range(B))
    print(f'[parity] configs[4] bs=8 s_step=250 (incl. capture): {sec:.1f} s = {sec / 250 * 1e3:.1f} ms per denoise step')
    assert whole.shape == (B, 6, 256, 256) and torch.isfinite(whole).all()
    assert float(whole.min()) >= 0 and float(whole.max()) <= 1
    for r in (0, 7):
        one, _ = run([r])
        assert torch.equal(one[0], whole[r]), r
    assert not torch.equal(whole[0], whole[7])
    d.rng.unkey()


def test_unet_ddp_fullsize_vs_oracle():
    """unconditional UNet (pixel-unshuffle Downsample, self-conditioning) at dim=64, 128x128"""
    m, sd = make_ddp(64, True)
    x, xs = rand((2, 3, 128, 128), 600), rand((2, 3, 128, 128), 601)
    t = torch.tensor([12, 907])
    with torch.no_grad():
        ref = OU.ddp_unet_forward(sd, x, t, xs, True)
    close_rel('ddp full', m(g(x), g(t), g(xs)).cpu(), ref, OUT_REL)


@pytest.mark.parametrize('tag', ['nosc', 'sc'])
def test_ddp_p_losses_forward_vs_golden(golden_dir, tag):
    """D9 for the unconditional class (DDP:772-820): the reference's own p_losses value for three objective / loss /
    p2-weight combinations and both outcomes of the self-conditioning draw (per-sample timesteps)"""
    from dmhomo_amd import ddpm
    gd = load(golden_dir, 'ddp_train_forward')
    sc = tag == 'sc'
    m, sd = make_ddp(8, sc)
    for obj, lt, gamma in (('pred_noise', 'l1', 0.), ('pred_x0', 'l2', 0.5), ('pred_v', 'l1', 1.0)):
        d = ddpm.GaussianDiffusion(m, image_size=16, timesteps=1000, objective=obj, loss_type=lt,
                                   p2_loss_weight_gamma=gamma).to(dev())
        for use in ((0, 1) if sc else (0,)):
            d._random = (lambda: 0.1) if use else (lambda: 0.9)
            loss = d.p_losses(g(T(gd['x_start'])), g(T(gd['t'])), noise=g(T(gd['noise'])))
            want = float(gd[f'{tag}.{obj}.{lt}.use{use}'])
            print(f'[parity] ddp p_losses {tag} {obj}/{lt} use_self_cond={use}: got {float(loss):.7f} want {want:.7f}')
            assert abs(float(loss) - want) <= 2e-5 * max(1.0, abs(want))
    val = d(g((T(gd['x_start']) + 1) / 2))                       # forward: normalise, random t (device RNG)
    assert val.ndim == 0 and bool(torch.isfinite(val))


def test_ddp_interpolate_vs_golden(golden_dir):
    """D10, DDP:737-754 against the reference: t = 0 is the reference's interpolate as it stands; t = 3 is the
    reference's own q_sample / p_sample in the chain the method means (it raises for t > 0: see oracle.diffusion.
    ddp_interpolate) — both from tests/golden/interpolate.npz, replaying the recorded draws"""
    from dmhomo_amd import ddpm
    gd = load(golden_dir, 'interpolate')
    m, sd = make_ddp(8, False)
    d = ddpm.GaussianDiffusion(m, image_size=16, timesteps=10, objective='pred_noise').to(dev())
    x1, x2 = g(T(gd['x1'])), g(T(gd['x2']))
    d.rng = ReplayDeviceRng([gd['t0.draw0'], gd['t0.draw1']])
    close('interpolate t=0', d.interpolate(x1, x2, t=0, lam=0.25).cpu(), T(gd['t0.out']), rtol=0, atol=2e-6)
    d.rng = ReplayDeviceRng([gd[f't3.draw{i}'] for i in range(4)])
    got = d.interpolate(x1, x2, t=3, lam=0.25)
    assert d.rng.i == 4
    close('interpolate t=3', got.cpu(), T(gd['t3.out']), rtol=0, atol=1e-4)
    # and against the oracle's restatement at a different blend / depth
    draws = [torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(80 + i)) for i in range(2 + 5)]
    with torch.no_grad():
        ref = OD.ddp_interpolate(sd, OD.schedule_buffers(10, 'cosine'), x1.cpu(), x2.cpu(), t=6, lam=0.7,
                                 objective='pred_noise', rng=OD.ReplayRng(draws))
    d.rng = ReplayDeviceRng(draws)
    close('interpolate t=6 vs oracle', d.interpolate(x1, x2, t=6, lam=0.7).cpu(), ref, rtol=0, atol=2e-4)


@pytest.mark.parametrize('tag,kw', [('learned', dict(learned_sinusoidal_cond=True)), ('random', dict(random_fourier_features=True)),
                                    ('learned8', dict(learned_sinusoidal_cond=True, learned_sinusoidal_dim=8))])
def test_unet_cfg_learned_sinusoidal_vs_golden(golden_dir, tag, kw):
    """Unet(learned_sinusoidal_cond=True / random_fourier_features=True) (RandomOrLearnedSinusoidalPosEmb, CFG:175-190,
    344-353): a bare Unet.forward caller gets the reference's numbers (GaussianDiffusion refuses such a model, CFG:514-515 —
    and so does ours); goldens from tests/golden/make_golden_r5.py"""
    from dmhomo_amd import cfg
    gd = load(golden_dir, 'r5')
    m, sd = make_cfg(8, **kw)
    assert 'time_mlp.0.weights' in sd and m.random_or_learned_sinusoidal_cond
    x, rf, mk, t = (g(T(gd['cfg.' + k])) for k in ('x', 'rf', 'mk', 't'))
    c = g(torch.zeros(x.shape[0], dtype=torch.long))
    close_rel(f'cfg {tag} keep', m(x, t, c, rf, mk, cond_drop_prob=0.).cpu(), T(gd[f'cfg.{tag}.keep']), OUT_REL)
    close_rel(f'cfg {tag} drop', m(x, t, c, rf, mk, cond_drop_prob=1.).cpu(), T(gd[f'cfg.{tag}.drop']), OUT_REL)
    with pytest.raises(AssertionError):
        cfg.GaussianDiffusion(m, image_size=32, timesteps=10)


def test_unet_ddp_learned_sinusoidal_vs_golden(golden_dir):
    gd = load(golden_dir, 'r5')
    from dmhomo_amd import ddpm
    m = ddpm.Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=3, learned_sinusoidal_cond=True)
    m.load_state_dict(det_state_dict(shapes_of(m), 1))
    m = m.to(dev())
    close_rel('ddp learned', m(g(T(gd['ddp.x'])), g(T(gd['cfg.t']))).cpu(), T(gd['ddp.learned']), OUT_REL)

"""-m gpu: end-to-end sampler parity on the DE-SATURATED weight set (tests/detweights.py, out_scale = DESAT).

With the plain deterministic weights the dim-64 UNet's output reaches |8..10|: 62-73 % of every pred_x0 step's x_start is
clamped to +-1 (CFG:612) and 2/3 of the final image is exactly 0 or 1 — those elements compare equal whatever the kernels
computed, so the S = 32 comparisons of test_gpu_unet.py see a third of their elements.  Here the
final projection is scaled so that < 5 % of x_start is clamped — the fraction is asserted and printed — and the same runs are
repeated: B = 2 against the oracle at full depth, rows 13 / 24 of the bs = 25 batch, plus two variants no full-size test
covered: objective = 'pred_noise', and clip_denoised = False through ddim_sample (no clamp at all: every element carries the
accumulated error of 32 steps).  Tolerances: the existing sampler bar (atol 2e-4 on the [0,1] image) where it holds; where a
variant amplifies fp32 rounding by construction the honest bound is written at the test with the measured number.
"""
import os

import numpy as np
import pytest
import torch

from gpu_util import dev, close, report, ReplayDeviceRng
from detweights import DESAT
from oracle import diffusion as OD
from test_gpu_unet import make_cfg, g, _fullsize_s32_inputs

pytestmark = pytest.mark.gpu


def _oracle(sd, B, rows=None, draws=None, **kw):
    rf01, flow, mk, c = _fullsize_s32_inputs(25 if rows else B)
    if rows:
        rf01, flow, mk, c = rf01[rows], flow[rows], mk[rows], c[rows]
    torch.manual_seed(99)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rec = OD.ReplayRng(draws) if draws is not None else OD.RecordRng()
    trace = []
    with torch.no_grad():
        ref, _, _ = OD.cfg_sample(sd, OD.schedule_buffers(1000, 'cosine'), c, rf01, flow, mk, image_size=128, channels=6,
                                  sampling_timesteps=32, rng=rec, trace=trace, **kw)
    return ref, trace, rec, (rf01, flow, mk, c)


def _clamped_fraction(trace, objective):
    """fraction of x_start elements sitting exactly on the clamp (CFG:612), over all steps"""
    n = sum(int(t['x_start'].numel()) for t in trace)
    return sum(int((t['x_start'].abs() == 1).sum()) for t in trace) / n


def test_sample_fullsize_s32_desaturated_vs_oracle():
    """configs[1]'s depth (dim 64, 128x128, T = 1000, s_step = 32, pred_x0, clip) on the de-saturated weights, B = 2"""
    from dmhomo_amd import cfg, ops
    m, sd = make_cfg(64, out_scale=DESAT)
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    ref, rtrace, rec, (rf01, flow, mk, c) = _oracle(sd, 2, objective='pred_x0')
    frac = _clamped_fraction(rtrace, 'pred_x0')
    edge = float(((ref == 0) | (ref == 1)).float().mean())
    print(f'[parity] de-saturated weights: clamped fraction of x_start over 32 steps = {frac:.4f}, final image on 0/1 = {edge:.4f}')
    assert 0 < frac < 0.05 and edge < 0.05
    d.rng = ReplayDeviceRng(rec.draws)
    trace = []
    img, _, _ = d._ddim_sample(g(c), ops.affine(g(rf01), 2., -1.), g(flow), g(mk), (2, 6, 128, 128), trace=trace)
    drift = [float((a['x_start'].cpu() - b['x_start']).abs().max()) for a, b in zip(trace, rtrace)]
    print('[parity] de-saturated S=32 full size: per-step max|x_start - oracle| = ' + ' '.join(f'{e:.1e}' for e in drift))
    close('sample S=32 full, de-saturated', img.cpu(), ref, rtol=0, atol=2e-4)
    u8 = ops.to_uint8(img).cpu().numpy().astype(np.int32)
    ru8 = (ref.numpy() * 255).astype(np.uint8).astype(np.int32)
    assert np.abs(u8 - ru8).max() <= 1


def test_sample_bs25_s32_desaturated_high_rows_vs_oracle():
    """rows 13 and 24 of the bs = 25 'streams' + captured-step + de-duplicated run — the bench's own configuration with every
    switch on — against the oracle on those two samples with the same draws, on the de-saturated weights"""
    from dmhomo_amd import cfg
    from dmhomo_amd.distributed import SampleIndexedRng
    m, sd = make_cfg(64, out_scale=DESAT)
    m.cfg_mode, m.dedup_dropped_rows = 'streams', True
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    rf01, flow, mk, c = _fullsize_s32_inputs(25)
    rows = [13, 24]
    # the draws of the keyed generator for those two rows: a second generator with the same key draws them in the sampler's
    # order (initial noise, then per step the class-dropout uniform and — except on the last step — the step noise)
    side = SampleIndexedRng(11, rows, dev())
    draws = [side.randn((2, 6, 128, 128), dev()).cpu()]
    for k in range(32):
        draws.append(side.uniform(2, dev()).cpu())
        if k < 31:
            draws.append(side.randn((2, 6, 128, 128), dev()).cpu())
    d.rng = SampleIndexedRng(11, range(25), dev())
    d.hip_graph = True
    img, _, _ = d.sample(g(c), g(rf01), g(flow), g(mk))
    d.hip_graph = False
    m.cfg_mode, m.dedup_dropped_rows = 'batched', False
    ref, rtrace, _, _ = _oracle(sd, 2, rows=rows, draws=draws, objective='pred_x0')
    frac = _clamped_fraction(rtrace, 'pred_x0')
    print(f'[parity] de-saturated bs=25 rows 13, 24: clamped fraction of x_start = {frac:.4f}')
    assert frac < 0.05
    close('sample bs=25 S=32 rows 13, 24 vs oracle, de-saturated, graph + streams + dedup', img[rows].cpu(), ref, rtol=0, atol=2e-4)


def test_sample_fullsize_s32_unclipped_vs_oracle():
    """ddim_sample(..., clip_denoised=False) (CFG:670) at full size and depth, de-saturated weights: nothing is clamped anywhere,
    every element of the result carries 32 steps of accumulated difference"""
    from dmhomo_amd import cfg, ops
    m, sd = make_cfg(64, out_scale=DESAT)
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev())
    ref, rtrace, rec, (rf01, flow, mk, c) = _oracle(sd, 2, objective='pred_x0', clip_denoised=False)
    assert _clamped_fraction(rtrace, 'pred_x0') < 1e-4          # (only a value that happens to be exactly +-1)
    d.rng = ReplayDeviceRng(rec.draws)
    img, _, _ = d.ddim_sample(g(c), ops.affine(g(rf01), 2., -1.), g(flow), g(mk), (2, 6, 128, 128), clip_denoised=False)
    assert float(ref.max()) > 1.0 or float(ref.min()) < 0.0     # the unclipped sample really leaves [0, 1]
    close('ddim_sample clip_denoised=False S=32 full, de-saturated', img.cpu(), ref, rtol=0, atol=2e-4)


def test_sample_fullsize_s32_pred_noise_vs_oracle():
    """objective = 'pred_noise' (CFG:614-617) at full size and depth: x_start = sqrt_recip_ac * x - sqrt_recipm1_ac * eps with
    sqrt_recip_ac[999] = 20291 — the first steps amplify the network's fp32 rounding (1e-6 of its output scale) by 1e4 before
    the clamp, so early x_start differ by up to ~1e-2 WHERE THEY ARE NOT CLAMPED; later steps (small factors) contract the
    difference again.  What is pinned: the final image at the sampler bar, and the per-step drift printed."""
    from dmhomo_amd import cfg, ops
    m, sd = make_cfg(64, out_scale=DESAT)
    d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_noise').to(dev())
    ref, rtrace, rec, (rf01, flow, mk, c) = _oracle(sd, 2, objective='pred_noise')
    frac = _clamped_fraction(rtrace, 'pred_noise')
    last = float((rtrace[-1]['x_start'].abs() == 1).float().mean())
    print(f'[parity] pred_noise: clamped fraction of x_start over 32 steps = {frac:.4f}, at the last step = {last:.4f}')
    d.rng = ReplayDeviceRng(rec.draws)
    trace = []
    img, _, _ = d._ddim_sample(g(c), ops.affine(g(rf01), 2., -1.), g(flow), g(mk), (2, 6, 128, 128), trace=trace)
    drift = [float((a['x_start'].cpu() - b['x_start']).abs().max()) for a, b in zip(trace, rtrace)]
    print('[parity] pred_noise S=32 full size: per-step max|x_start - oracle| = ' + ' '.join(f'{e:.1e}' for e in drift))
    report('sample S=32 full, pred_noise', img.cpu(), ref)
    close('sample S=32 full, pred_noise', img.cpu(), ref, rtol=0, atol=2e-4)

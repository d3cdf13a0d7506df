"""-m gpu: the surface members added in round 4 (predict_* / q_posterior / p_mean_variance, model_predictions
with a timestep per row, forward_with_cond_scale(*args, cond_scale, **kwargs), norm_grid / get_grid / DLT_solve /
get_flow_np) against outputs of the REFERENCE (tests/golden/surface.npz, made by make_golden_r4.py).

Bars: elementwise affine combinations in the reference's op order: bit-exact (torch.equal); anything behind a UNet forward:
2e-5 of the tensor's scale (the UNet bar); norm_grid / get_grid: bit-exact; get_flow_np (float64): 1e-12 abs; DLT_solve:
1e-10 of max |H| (measured 2e-13; the reference takes a pseudo-inverse, the kernels solve the square / normal-equation
system)."""
import os

import numpy as np
import pytest
import torch

from gpu_util import dev, close_rel, ReplayDeviceRng
from test_gpu_unet import make_cfg, make_ddp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gd(golden_dir):
    return {k: v for k, v in np.load(os.path.join(golden_dir, 'surface.npz')).items()}


def G(a):
    return torch.from_numpy(np.asarray(a)).to(dev())


def test_affine_combinations_per_row_bit_exact(gd):
    from dmhomo_amd import cfg, ddpm
    m, _ = make_cfg(8)
    mu, _ = make_ddp(8, False)
    x, other, t = G(gd['x']), G(gd['other']), G(gd['t'])
    for d in (cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4).to(dev()),
              ddpm.GaussianDiffusion(mu, image_size=16, timesteps=1000, sampling_timesteps=4).to(dev())):
        for name, args in (('predict_start_from_noise', (x, t, other)), ('predict_noise_from_start', (x, t, other)),
                           ('predict_v', (x, t, other)), ('predict_start_from_v', (x, t, other))):
            got = getattr(d, name)(*args).cpu()
            assert torch.equal(got, torch.from_numpy(gd[f'cfg.{name}'])), name
        mean, var, logvar = d.q_posterior(other, x, t)
        assert torch.equal(mean.cpu(), torch.from_numpy(gd['cfg.q_posterior.mean']))
        assert var.shape == (3, 1, 1, 1) and torch.equal(var.cpu(), torch.from_numpy(gd['cfg.q_posterior.var']))
        assert torch.equal(logvar.cpu(), torch.from_numpy(gd['cfg.q_posterior.logvar']))


@pytest.mark.parametrize('obj', ['pred_noise', 'pred_x0', 'pred_v'])
def test_cfg_model_predictions_per_row_timesteps(gd, obj):
    from dmhomo_amd import cfg
    m, _ = make_cfg(8)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective=obj).to(dev())
    x, t, rf, mk = G(gd['x']), G(gd['t']), G(gd['rgb_flow']), G(gd['mask'])
    classes = torch.zeros(3, dtype=torch.long, device=dev())
    for clip in (0, 1):
        d.rng = ReplayDeviceRng([gd[f'cfg.mp.{obj}.{clip}.draw']])
        mp = d.model_predictions(x, t, classes, rf, mk, cond_scale=3., clip_x_start=bool(clip))
        # pred_noise at t = 999 divides by sqrt_recipm1 ~ 2e4 differences of O(1) numbers: judged on the tensor's scale;
        # the clamped x_start on the scale of the unclamped one (the clamp cuts the scale, not the error of what it keeps)
        want = torch.from_numpy(gd[f'cfg.mp.{obj}.{clip}.x_start'])
        scale = float(torch.from_numpy(gd[f'cfg.mp.{obj}.0.x_start']).abs().max())
        err = float((mp.pred_x_start.cpu() - want).abs().max())
        print(f'[parity] cfg model_predictions {obj} clip={clip} x_start: max_abs={err:.3e} = {err / scale:.3e} of the unclamped scale')
        assert err <= 2e-5 * scale
        close_rel(f'cfg model_predictions {obj} clip={clip} pred_noise', mp.pred_noise, torch.from_numpy(gd[f'cfg.mp.{obj}.{clip}.pred_noise']), 2e-5)
    # rows of a uniform-timestep batch: the fused one-pass path and the row-by-row path agree bit for bit
    tu = torch.full((3,), 500, device=dev(), dtype=torch.long)
    d.rng = ReplayDeviceRng([gd[f'cfg.mp.{obj}.1.draw']] * 2)
    a = d.model_predictions(x, tu, classes, rf, mk, cond_scale=3., clip_x_start=True)
    cond, null, computed = d._network(x, tu, classes, rf, mk, 3.)
    assert computed is None                               # (dedup_dropped_rows is off: every conditional row was computed)
    from dmhomo_amd import ops
    from dmhomo_amd._lib import DmhStep
    blend = DmhStep(objective=1, clip=0, mode=ops.MODE_LAST, cond_scale=3., sqrt_recip_ac=1., sqrt_recipm1_ac=1.)
    out, _, _ = ops.sampler_step(blend, cond, null, cond, None, want_x_start=False)
    b = d._predictions_per_row(out, x, tu, True)
    assert torch.equal(a.pred_x_start, b.pred_x_start) and torch.equal(a.pred_noise, b.pred_noise)


@pytest.mark.parametrize('tag', ['nosc', 'sc'])
def test_ddp_model_predictions_and_p_mean_variance(gd, tag):
    from dmhomo_amd import ddpm
    mu, _ = make_ddp(8, tag == 'sc')
    x, t = G(gd[f'ddp.{tag}.x']), G(gd['t'])
    xsc = G(gd[f'ddp.{tag}.x_self_cond']) if tag == 'sc' else None
    for obj in ('pred_noise', 'pred_x0', 'pred_v'):
        d = ddpm.GaussianDiffusion(mu, image_size=16, timesteps=1000, sampling_timesteps=4, objective=obj).to(dev())
        mp = d.model_predictions(x, t, xsc, clip_x_start=True)
        close_rel(f'ddp {tag} {obj} x_start', mp.pred_x_start, torch.from_numpy(gd[f'ddp.{tag}.mp.{obj}.x_start']), 2e-5)
        close_rel(f'ddp {tag} {obj} pred_noise', mp.pred_noise, torch.from_numpy(gd[f'ddp.{tag}.mp.{obj}.pred_noise']), 2e-5)
        mean, var, logvar, xs = d.p_mean_variance(x, t, xsc, clip_denoised=True)
        close_rel(f'ddp {tag} {obj} p_mean_variance mean', mean, torch.from_numpy(gd[f'ddp.{tag}.pmv.{obj}.mean']), 2e-5)
        close_rel(f'ddp {tag} {obj} p_mean_variance x_start', xs, torch.from_numpy(gd[f'ddp.{tag}.pmv.{obj}.x_start']), 2e-5)
        assert torch.equal(var.cpu(), torch.from_numpy(gd[f'ddp.{tag}.pmv.{obj}.var']))
        assert torch.equal(logvar.cpu(), torch.from_numpy(gd[f'ddp.{tag}.pmv.{obj}.logvar']))


def test_forward_with_cond_scale_signature_semantics(gd):
    m, _ = make_cfg(8)
    x, t, rf, mk = G(gd['x']), G(gd['t']), G(gd['rgb_flow']), G(gd['mask'])
    classes = torch.zeros(3, dtype=torch.long, device=dev())
    a = m.forward_with_cond_scale(x, t, classes, rf, mk, cond_scale=1., cond_drop_prob=0.)
    close_rel('forward_with_cond_scale(cond_scale=1, cond_drop_prob=0)', a, torch.from_numpy(gd['cfg.fwcs.scale1.drop0']), 2e-5)
    b = m.forward_with_cond_scale(x, t, classes, rgb_flow=rf, mask=mk, cond_drop_prob=1.)
    close_rel('forward_with_cond_scale(cond_drop_prob=1)', b, torch.from_numpy(gd['cfg.fwcs.scale1.drop1']), 2e-5)
    with pytest.raises(TypeError, match="multiple values for keyword argument 'cond_drop_prob'"):
        m.forward_with_cond_scale(x, t, classes, rf, mk, cond_scale=3., cond_drop_prob=0.)
    # keyword form with a guidance scale: null + (cond - null) * s of the two passes
    m.rng = ReplayDeviceRng([torch.full((3,), 0.25)])                       # keep every class
    c = m.forward_with_cond_scale(x, t, classes, rgb_flow=rf, mask=mk, cond_scale=2.)
    want = b + (a - b) * 2.
    close_rel('forward_with_cond_scale(cond_scale=2) vs its two passes', c, want.cpu(), 2e-6)


def test_cfg_ancestral_methods_raise_like_the_reference(gd):
    from dmhomo_amd import cfg
    m, _ = make_cfg(8)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective='pred_x0').to(dev())
    x, t = G(gd['x']), G(gd['t'])
    classes = torch.zeros(3, dtype=torch.long, device=dev())
    with pytest.raises(TypeError, match="missing 1 required positional argument: 'mask'"):
        d.p_mean_variance(x, t, classes, 3.)
    with pytest.raises(TypeError, match="missing 1 required positional argument: 'mask'"):
        d.p_sample(x, 5, classes)


def test_grid_helpers_bit_exact(gd):
    from dmhomo_amd.denoising_diffusion_models import denoising_diffusion_pytorch as ddp
    ng = ddp.norm_grid(G(gd['norm_grid.in']))
    assert ng.shape == (2, 9, 13, 2) and torch.equal(ng.cpu(), torch.from_numpy(gd['norm_grid.out']))
    gg = ddp.get_grid(2, 5, 7, start=3)
    assert gg.is_cuda and gg.dtype == torch.float32 and torch.equal(gg.cpu(), torch.from_numpy(gd['get_grid']))


def test_get_flow_np_bands(gd):
    from dmhomo_amd.denoising_diffusion_models import denoising_diffusion_pytorch as ddp
    Hs, idx = gd['get_flow_np.H'], gd['get_flow_np.idx']
    for b in range(2):
        fl = ddp.get_flow_np(Hs[b:b + 1], idx, image_size_h=20, image_size_w=16)
        assert fl.shape == (20, 16, 2) and fl.dtype == np.float64
        err = np.abs(fl - gd['get_flow_np.flow'][b]).max()
        print(f'[parity] get_flow_np divide=3 batch {b}: max_abs={err:.3e}')
        assert err <= 1e-12
    with pytest.raises(ValueError):          # the reference's squeeze().transpose(1, 2, 0) cannot take a batch either
        ddp.get_flow_np(Hs, idx, image_size_h=20, image_size_w=16)
    one = ddp.homo_to_flow(Hs[0, 0][None, None], H=20, W=16)
    assert one.dtype == np.float32 and one.shape == (20, 16, 2)


def test_dlt_solve_mesh_and_points(gd):
    from dmhomo_amd.denoising_diffusion_models import denoising_diffusion_pytorch as ddp
    for name in ('mesh2', 'mesh1', 'points'):
        src, off, want = (torch.from_numpy(gd[f'dlt.{name}.{k}']) for k in ('src', 'off', 'H'))
        got = ddp.DLT_solve(src.to(dev()), off.to(dev()))
        assert got.shape == want.shape and got.dtype == torch.float64
        close_rel(f'DLT_solve {name}', got, want, 1e-10)             # measured 2.1e-13
        got_cpu_in = ddp.DLT_solve(src, off)                    # host tensors are moved for the solve; the result comes back
        assert got_cpu_in.device.type == 'cpu' and torch.equal(got_cpu_in, got.cpu())   # where the reference's would be (DDP:1641)
    # dtype follows the reference's torch.cat((h8, ones)): float32 points give a float32 result
    src, off = (torch.from_numpy(gd[f'dlt.mesh1.{k}']).float() for k in ('src', 'off'))
    assert ddp.DLT_solve(src.to(dev()), off.to(dev())).dtype == torch.float32


def test_dlt_solve_degenerate_corners_stay_finite():
    """a 4-point cell with a repeated corner (rank-deficient 8x8 system): the reference's torch.linalg.pinv (DDP:1639) returns
    the finite minimum-norm solution; the direct solve has no pivot there and goes through the regularised normal equations
    — finite, and the minimum-norm solution to ~1e-6 (asserted at 2e-5)"""
    from dmhomo_amd.denoising_diffusion_models import denoising_diffusion_pytorch as ddp
    src = torch.tensor([[[[0., 0.], [1., 0.], [1., 0.], [0., 1.]]]], dtype=torch.float64)     # corner 1 twice
    off = torch.tensor([[[[.1, .05], [.02, -.03], [.02, -.03], [-.04, .01]]]], dtype=torch.float64)
    got = ddp.DLT_solve(src.to(dev()), off.to(dev())).cpu()
    assert torch.isfinite(got).all()
    dst = src + off
    rows = []
    for (x, y), (u, v) in zip(src[0, 0].tolist(), dst[0, 0].tolist()):
        rows += [[x, y, 1, 0, 0, 0, -u * x, -u * y], [0, 0, 0, x, y, 1, -v * x, -v * y]]
    A, b = torch.tensor(rows, dtype=torch.float64), dst.reshape(8, 1)
    want = torch.cat([(torch.linalg.pinv(A) @ b).flatten(), torch.ones(1, dtype=torch.float64)]).reshape(3, 3)
    assert float((got[0, 0] - want).abs().max()) < 2e-5, (got, want)
    # a well-posed cell beside it is untouched by the guard (exact 4-point homography)
    src2 = torch.tensor([[[[0., 0.], [1., 0.], [1., 1.], [0., 1.]]]], dtype=torch.float64)
    off2 = torch.tensor([[[[.1, .05], [.02, -.03], [.07, .02], [-.04, .01]]]], dtype=torch.float64)
    H = ddp.DLT_solve(src2.to(dev()), off2.to(dev())).cpu()[0, 0]
    p = torch.cat([src2[0, 0], torch.ones(4, 1, dtype=torch.float64)], 1) @ H.T
    assert float((p[:, :2] / p[:, 2:] - (src2 + off2)[0, 0]).abs().max()) < 1e-12

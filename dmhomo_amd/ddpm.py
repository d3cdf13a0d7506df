"""Unconditional UNet + GaussianDiffusion + Trainer + geometry helpers on MI355X.

Host-side mirror of ``DGM/denoising_diffusion_models/denoising_diffusion_pytorch.py`` (tag DDP):
``Unet`` DDP:315-447, ``GaussianDiffusion`` DDP:481-817 (``p_sample`` / ``p_sample_loop`` /
``ddim_sample`` are the functions BASELINE.json's north_star names), the condition-builder
helpers DDP:913-988,1262-1299,1471-1486,1558-1678 and the ``Trainer`` surface DDP:1681-2021.
All tensor values come from libdmhomo_hip.so; there is no CPU path.
"""
import os
from pathlib import Path

import numpy as np
import torch
from torch import nn

from . import _params as P
from . import ops
from ._lib import DmhStep
from .cfg import DeviceRng, ModelPrediction, ScheduleHost, default, exists, extract  # noqa: F401
from .engine import UnetEngine
from .schedule import make_buffers, ddim_pairs, linear_beta_schedule, cosine_beta_schedule  # noqa: F401

__version__ = '0.1.0'


class Unet(nn.Module):
    """DDP:315-447 (pixel-unshuffle Downsample, optional self-conditioning, time embedding only)."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16):
        super().__init__()
        self.channels = channels
        self.self_condition = self_condition
        input_channels = channels * (2 if self_condition else 1)
        init_dim = default(init_dim, dim)
        self.init_conv = nn.Conv2d(input_channels, init_dim, 7, padding=3)
        time_dim = dim * 4
        self.random_or_learned_sinusoidal_cond = learned_sinusoidal_cond or random_fourier_features
        # (GaussianDiffusion refuses such a model — CFG:514-515 — so it serves bare Unet.forward callers only)
        pos_emb, fourier_dim = P.time_embedding(dim, learned_sinusoidal_cond, random_fourier_features, learned_sinusoidal_dim)
        self.time_mlp = nn.Sequential(pos_emb, nn.Linear(fourier_dim, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim))
        self.out_dim = default(out_dim, channels * (1 if not learned_variance else 2))
        P.build_trunk(self, dim, init_dim, dim_mults, input_channels, time_dim, resnet_block_groups, self.out_dim,
                      P.downsample_ddp)
        self._engine = UnetEngine(self, groups=resnet_block_groups)

    def forward(self, x, time, x_self_cond=None):
        if not x.is_cuda:
            raise RuntimeError('dmhomo_amd.Unet runs on the GPU only (HIP kernels); move the inputs with .cuda()')
        eng = self._engine
        eng.ensure_prepared()
        x = x.to(torch.float32).contiguous()
        time = time.to(torch.int64).contiguous()
        if self.self_condition:
            sc = default(x_self_cond, lambda: torch.zeros_like(x)).to(torch.float32).contiguous()
            xin = ops.assemble_input(sc, x, None, cpad=eng.cin_pad)          # cat((x_self_cond, x)), DDP:411
        else:
            xin = ops.assemble_input(x, None, None, cpad=eng.cin_pad)
        cond = eng.embed(time, None, 1)
        return eng.trunk(eng.stem(xin), cond)


class GaussianDiffusion(nn.Module, ScheduleHost):
    """DDP:481-817, sampling side."""

    def __init__(self, model, *, image_size, timesteps=1000, sampling_timesteps=None, loss_type='l1',
                 objective='pred_noise', beta_schedule='cosine', p2_loss_weight_gamma=0., p2_loss_weight_k=1,
                 ddim_sampling_eta=1.):
        super().__init__()
        assert not (type(self) == GaussianDiffusion and model.channels != model.out_dim)
        assert not model.random_or_learned_sinusoidal_cond
        self.model = model
        self.channels = self.model.channels
        self.self_condition = self.model.self_condition
        self.image_size = image_size
        self.objective = objective
        assert objective in {'pred_noise', 'pred_x0', 'pred_v'}, \
            'objective must be either pred_noise (predict noise) or pred_x0 (predict image start) or pred_v (predict v)'
        bufs = make_buffers(beta_schedule, timesteps, p2_loss_weight_gamma, p2_loss_weight_k)
        self.num_timesteps = int(bufs['betas'].shape[0])
        self.loss_type = loss_type
        self.sampling_timesteps = default(sampling_timesteps, timesteps)
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        for name, val in bufs.items():
            self.register_buffer(name, val)
        self.rng = DeviceRng()

    def _step(self, host, t, mode, clip, c=(0., 0., 0.)):
        return DmhStep(objective=ops.OBJECTIVE[self.objective], clip=int(bool(clip)), mode=mode, cond_scale=1.,
                       sqrt_recip_ac=float(host['sqrt_recip_alphas_cumprod'][t]),
                       sqrt_recipm1_ac=float(host['sqrt_recipm1_alphas_cumprod'][t]),
                       sqrt_ac=float(host['sqrt_alphas_cumprod'][t]),
                       sqrt_1m_ac=float(host['sqrt_one_minus_alphas_cumprod'][t]),
                       c0=float(c[0]), c1=float(c[1]), c2=float(c[2]))

    def model_predictions(self, x, t, x_self_cond=None, clip_x_start=False):
        """DDP:613-634 (one timestep per batch, as every sampler uses it: one fused pass; a timestep per row: row by row)."""
        out = self.model(x, t, x_self_cond)
        t0 = self._uniform_time(t)
        if t0 is None:
            return self._predictions_per_row(out, x.contiguous(), t, clip_x_start)
        step = self._step(self._host(), t0, ops.MODE_LAST, clip_x_start)
        _, x_start, pred_noise = ops.sampler_step(step, out, None, x.contiguous(), None, True, True)
        return ModelPrediction(pred_noise, x_start)

    @staticmethod
    def _uniform_time(t):
        t0 = int(t[0])
        return None if (t.numel() > 1 and not bool((t == t0).all())) else t0

    def p_mean_variance(self, x, t, x_self_cond=None, clip_denoised=True):
        """DDP:636-645: (posterior mean, variance, clipped log variance, x_start) of one ancestral step."""
        preds = self.model_predictions(x, t, x_self_cond)
        x_start = preds.pred_x_start
        if clip_denoised:
            x_start = ops.rows_lincomb(x_start, torch.ones((x.shape[0],), device=x.device, dtype=torch.float32),
                                       clamp=True)
        model_mean, posterior_variance, posterior_log_variance = self.q_posterior(x_start=x_start, x_t=x, t=t)
        return model_mean, posterior_variance, posterior_log_variance, x_start

    @torch.no_grad()
    def p_sample(self, x, t: int, x_self_cond=None, clip_denoised=True):
        """DDP:647-661: one ancestral step -> (pred_img, x_start)."""
        host = self._host()                                   # (cached per buffer version)
        bt = torch.full((x.shape[0],), t, device=x.device, dtype=torch.long)
        out = self.model(x, bt, x_self_cond)
        # exp(0.5 * logvar) as fp32 0-dim tensor arithmetic, DDP:660
        sd = float((0.5 * host['posterior_log_variance_clipped'][t]).exp())
        step = self._step(host, t, ops.MODE_DDPM, clip_denoised,
                          (host['posterior_mean_coef1'][t], host['posterior_mean_coef2'][t], sd))
        noise = self.rng.randn(x.shape, x.device).contiguous() if t > 0 else None
        img, x_start, _ = ops.sampler_step(step, out, None, x.contiguous(), noise, True)
        return img, x_start

    @torch.no_grad()
    def p_sample_loop(self, shape):
        """DDP:663-680."""
        device = self.betas.device
        host = self._host()
        img = self.rng.randn(shape, device).contiguous()
        x_start = None
        for t in reversed(range(0, self.num_timesteps)):
            self_cond = x_start if self.self_condition else None
            img, x_start = self.p_sample(img, t, self_cond)
        img = ops.affine(img, 0.5, 0.5)
        ops.affine_tail_(img, img.shape[1] - 2, 2., -1.)          # flow channels back to [-1, 1], DDP:679
        return img

    @torch.no_grad()
    def ddim_sample(self, shape, clip_denoised=True):
        """DDP:682-729."""
        batch, device = shape[0], self.betas.device
        host = self._host()
        img = self.rng.randn(shape, device).contiguous()
        x_start = None
        for time, time_next in ddim_pairs(self.num_timesteps, self.sampling_timesteps):
            time_cond = torch.full((batch,), time, device=device, dtype=torch.long)
            self_cond = x_start if self.self_condition else None
            out = self.model(img, time_cond, self_cond)
            if time_next < 0:
                step, noise = self._step(host, time, ops.MODE_LAST, clip_denoised), None
            else:
                step = self._step(host, time, ops.MODE_DDIM, clip_denoised, self._ddim_coef(host, time, time_next))
                noise = self.rng.randn(shape, device).contiguous()
            img, x_start, _ = ops.sampler_step(step, out, None, img, noise, True)
        img = ops.affine(img, 0.5, 0.5)
        ops.affine_tail_(img, img.shape[1] - 2, 2. * 512, -512.)   # (x*2-1)*512, DDP:728
        return img

    @torch.no_grad()
    def sample(self, batch_size=16):
        """DDP:731-735."""
        shape = (batch_size, self.channels, self.image_size, self.image_size)
        return self.ddim_sample(shape) if self.is_ddim_sampling else self.p_sample_loop(shape)

    @torch.no_grad()
    def interpolate(self, x1, x2, t=None, lam=0.5):
        """DDP:737-754: q_sample both images at step t, blend, denoise with p_sample down to 0.
        The reference assigns p_sample's (pred_img, x_start) TUPLE back to ``img`` (DDP:750-752 vs 661), so its loop
        cannot run past the first step; the evident intent — carry pred_img — is implemented."""
        b = x1.shape[0]
        t = default(t, self.num_timesteps - 1)
        assert x1.shape == x2.shape
        t_batched = torch.full((b,), t, device=x1.device, dtype=torch.long)
        xt1, xt2 = self.q_sample(x1, t_batched), self.q_sample(x2, t_batched)
        img = ops.lerp(xt1, xt2, float(lam))                               # (1 - lam) * xt1 + lam * xt2
        host = self._host()
        for i in reversed(range(0, t)):
            img, _ = self.p_sample(img, i)
        return img

    def q_sample(self, x_start, t, noise=None):
        """DDP:756-761."""
        noise = default(noise, lambda: self.rng.randn(x_start.shape, x_start.device))
        ca = self.sqrt_alphas_cumprod.gather(-1, t).contiguous()
        cb = self.sqrt_one_minus_alphas_cumprod.gather(-1, t).contiguous()
        return ops.q_sample(x_start.contiguous(), noise.contiguous(), ca, cb)

    @property
    def loss_fn(self):
        """DDP:763-770 — the name of the elementwise loss (the reduction runs in dmh_diff_mean)."""
        if self.loss_type in ('l1', 'l2'):
            return self.loss_type
        raise ValueError(f'invalid loss type {self.loss_type}')

    _random = staticmethod(__import__('random').random)      # the `random() < 0.5` self-conditioning draw of DDP:785

    @torch.no_grad()
    def p_losses(self, x_start, t, noise=None):
        """DDP:772-811, FORWARD VALUE: q_sample -> (self-conditioning pass, half of the time) -> UNet -> per-sample L1 / L2
        mean -> x p2_loss_weight[t] -> mean.  The DGM trains the conditional class (classifier_free_guidance, built with
        its backward in dmhomo_amd.train); this unconditional twin returns the loss value without an autograd graph."""
        squared = self.loss_fn == 'l2'
        x_start = x_start.to(torch.float32).contiguous()
        noise = default(noise, lambda: self.rng.randn(x_start.shape, x_start.device)).to(torch.float32).contiguous()
        t = t.to(torch.int64).contiguous()
        x = self.q_sample(x_start, t, noise)
        x_self_cond = None
        if self.self_condition and self._random() < 0.5:
            out0 = self.model(x, t, None)                    # model_predictions(x, t).pred_x_start, DDP:787 (no clamp)
            if self.objective == 'pred_x0':
                x_self_cond = out0
            else:                                            # per-sample coefficients: t differs from row to row
                ca = (self.sqrt_recip_alphas_cumprod if self.objective == 'pred_noise' else self.sqrt_alphas_cumprod)
                cb = (self.sqrt_recipm1_alphas_cumprod if self.objective == 'pred_noise'
                      else self.sqrt_one_minus_alphas_cumprod)
                x_self_cond = ops.q_sample(x, out0.contiguous(), ca.gather(-1, t).contiguous(),
                                           (-cb).gather(-1, t).contiguous())
        model_out = self.model(x, t, x_self_cond)
        if self.objective == 'pred_noise':
            target = noise
        elif self.objective == 'pred_x0':
            target = x_start
        elif self.objective == 'pred_v':                     # predict_v, DDP:596-598
            target = ops.q_sample(noise, x_start, self.sqrt_alphas_cumprod.gather(-1, t).contiguous(),
                                  (-self.sqrt_one_minus_alphas_cumprod).gather(-1, t).contiguous())
        else:
            raise ValueError(f'unknown objective {self.objective}')
        loss = ops.diff_mean(model_out, target, None, squared)                         # (B,) per-sample means
        w = self.p2_loss_weight.gather(-1, t).contiguous()
        return ops.loss_combine(torch.zeros_like(loss), loss, w)                       # mean(loss * w), DDP:810-811

    def forward(self, img, *args, **kwargs):
        """DDP:813-820."""
        b, c, h, w = img.shape
        assert h == self.image_size and w == self.image_size, f'height and width of image must be {self.image_size}'
        t = torch.randint(0, self.num_timesteps, (b,), device=img.device).long()
        return self.p_losses(ops.affine(img.to(torch.float32), 2., -1.), t, *args, **kwargs)


# =====================================================================================
# condition builder & geometry (DDP:913-988, 1262-1299, 1471-1486, 1558-1678)
# =====================================================================================
def adapt_homography_to_preprocessing_v3(h0, w0, H, h1, w1):
    """G1, DDP:978-988: rescale a 3x3 homography between image sizes (host, float64, 27 flops)."""
    def m(h, w):
        return np.array([[w / 2.0, 0., w / 2.0], [0., h / 2.0, h / 2.0], [0., 0., 1.]])
    M0, M1 = m(h0, w0), m(h1, w1)
    return M1 @ (np.linalg.inv(M0) @ H @ M0) @ np.linalg.inv(M1)


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('dmhomo_amd geometry kernels need a GPU; there is no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


def homo_to_flow_rgb(homos, H, W, max_flow=256):
    """G2+G3 batched on device: homos (B,3,3) f64 (numpy or tensor) -> flow (B,2,H,W), rgb (B,3,H,W) fp32."""
    Hm = torch.as_tensor(np.asarray(homos) if not torch.is_tensor(homos) else homos, dtype=torch.float64)
    Hm = Hm.reshape(-1, 3, 3).to(_dev()).contiguous()
    return ops.homography_flow(Hm, H, W, max(max_flow, 1.))


def homo_to_flow(homo, H=600, W=800):
    """DDP:972-975 signature: one homography (any shape with 9 elements) -> (H, W, 2) float32 numpy."""
    flow, _ = ops.homography_flow(torch.as_tensor(np.asarray(homo), dtype=torch.float64).reshape(1, 3, 3)
                                  .to(_dev()).contiguous(), H, W, 256., want_rgb=False)
    return flow[0].permute(1, 2, 0).contiguous().cpu().numpy()


def mesh_grid_np(B, H, W):
    """DDP:913-924: (B,3,H,W) integer homogeneous pixel coordinates (x, y, 1) (host numpy, as in the reference)."""
    x_base = np.tile(np.arange(0, W), (B, H, 1))
    y_base = np.tile(np.arange(0, H), (B, W, 1)).transpose(0, 2, 1)
    return np.stack([x_base, y_base, np.ones_like(x_base)], 1)


def get_flow_np(H_mat_mul, patch_indices, image_size_h=600, image_size_w=800):
    """DDP:927-969: ``divide`` homographies (batch, divide, 3, 3)-shaped, one per horizontal band of the image, applied to
    the homogeneous coordinates ``patch_indices`` (B,3,H,W) in float64 (w' + 1e-6, DDP:958-959) -> flow, squeezed and
    laid out (H, W, 2) as the reference returns it (dmh_homography_flow_points)."""
    H_mat_mul = np.asarray(H_mat_mul, dtype=np.float64)
    batch_size, divide = H_mat_mul.shape[0], H_mat_mul.shape[1]
    Hm = torch.from_numpy(np.ascontiguousarray(H_mat_mul.reshape(batch_size, divide, 3, 3))).to(_dev())
    idx = torch.from_numpy(np.ascontiguousarray(np.asarray(patch_indices, dtype=np.float64))).to(_dev())
    assert idx.shape[2:] == (image_size_h, image_size_w), (idx.shape, image_size_h, image_size_w)
    flow = ops.homography_flow_points(Hm, idx).cpu().numpy()
    return flow.squeeze().transpose(1, 2, 0)


def mesh_grid(B, H, W):
    """DDP:1283-1289: (B,2,H,W) int64 pixel coordinates (x, y) — on the host, like the reference's (its caller moves it
    with ``.type_as(x)``; the product's flow_warp never materialises it)."""
    x_base = torch.arange(0, W).repeat(B, H, 1)
    y_base = torch.arange(0, H).repeat(B, W, 1).transpose(1, 2)
    return torch.stack([x_base, y_base], 1)


def norm_grid(v_grid):
    """DDP:1292-1299: pixel coordinates (B,2,H,W) -> grid_sample coordinates (B,H,W,2) in [-1, 1] (dmh_norm_grid)."""
    return ops.norm_grid(v_grid.to(torch.float32))


def get_grid(batch_size, H, W, start=0):
    """DDP:1558-1574: (B,2,H,W) fp32 pixel coordinates + start, on the GPU (dmh_pixel_grid)."""
    return ops.pixel_grid(batch_size, H, W, start, _dev())


def DLT_solve(src_p, off_set):
    """DDP:1577-1644: src_p, off_set (bs, n, P, 2) -> (bs, n', 3, 3) homographies.  With a (divide+1)^2 mesh of points
    (n = 2 (divide+1)^2 coordinates laid out as the reference's index arithmetic expects) every mesh cell's four corners
    are gathered (tensor indexing) into its own 4-point system; with n == 1 (homo_gen) the P points are one least-squares
    system.  Solved in float64 by dmh_dlt_points."""
    bs = src_p.shape[0]
    divide = int(np.sqrt(len(src_p[0]) / 2) - 1)
    row_num = (divide + 1) * 2
    src_ps, off_sets = src_p, off_set
    cells = []
    for i in range(divide):
        for j in range(divide):
            k = 2 * j + row_num * i
            cells.append([k, k + 1, k + 2, k + 3, k + 2 + row_num, k + 3 + row_num, k + row_num, k + row_num + 1])
    if cells:
        src_ps = torch.cat([src_p[:, c].reshape(bs, 1, 4, 2) for c in cells], dim=1)
        off_sets = torch.cat([off_set[:, c].reshape(bs, 1, 4, 2) for c in cells], dim=1)
    bs, n, h, w = src_ps.shape
    dev = _dev()
    src64 = src_ps.reshape(bs * n, h, w).to(device=dev, dtype=torch.float64)
    off64 = off_sets.reshape(bs * n, h, w).to(device=dev, dtype=torch.float64)
    # solved in float64 on the GPU; handed back as the reference's result would be: on the inputs' device, in the dtype its
    # torch.cat((h8, ones)) promotes to (DDP:1641 — float64 for homo_gen's float64 grid, float32 for float32 points)
    out_dtype = torch.promote_types(torch.promote_types(src_p.dtype, off_set.dtype), torch.float32)
    return ops.dlt_points(src64, off64).reshape(bs, n, 3, 3).to(device=src_p.device, dtype=out_dtype)


def flow_warp(x, flow12, pad='border', mode='bilinear'):
    """G4, DDP:1262-1280: ``pad`` / ``mode`` are grid_sample's padding_mode / mode (DDP:1270-1274): 'border' | 'zeros' |
    'reflection' and 'bilinear' | 'nearest' | 'bicubic' — every value grid_sample takes for 4-D inputs (no caller in DGM
    passes anything but the defaults)."""
    if pad not in ops.FLOW_WARP_PAD or mode not in ops.FLOW_WARP_MODE:
        raise NotImplementedError(f"flow_warp(pad={pad!r}, mode={mode!r}): built are pad in {sorted(ops.FLOW_WARP_PAD)} and mode "
                                  f"in {sorted(ops.FLOW_WARP_MODE)} (DDP:1262-1280 forwards both to F.grid_sample)")
    return ops.flow_warp(x.to(torch.float32).contiguous(), flow12.to(torch.float32).contiguous(), pad=pad, mode=mode)


def homo_gen(flow):
    """G5, DDP:1647-1661: flow (B,2,H,W) -> (B,1,3,3) float64."""
    return ops.dlt_homography(flow.to(torch.float32).contiguous()).reshape(-1, 1, 3, 3)


def saveTrainPair(torch_tensor, mask, flows):
    """G6, DDP:1664-1678: {"imgs": uint8 (B,6,H,W), "homos": float64 (B,3,3)}."""
    assert torch.max(torch_tensor) <= 1, \
        f'image should be normalized to [0, 1], not[{torch.min(torch_tensor)}, {torch.max(torch_tensor)}]'
    imgs = ops.to_uint8(torch_tensor.detach().to(torch.float32).contiguous())
    homos = homo_gen(flows)
    return {'imgs': imgs.cpu().numpy(), 'homos': homos.cpu().numpy().squeeze()}


# =====================================================================================
# Trainer surface (DDP:1681-2021): constructor, save / load, sample
# =====================================================================================
def _clone_without_engine_caches(model):
    """deepcopy of a module tree whose Unets get fresh (empty) kernel-program caches instead of copies"""
    import copy
    memo, engines = {}, []
    for mod in model.modules():
        eng = mod.__dict__.get('_engine')
        if eng is not None:
            memo[id(eng)] = None
            engines.append(eng)
        # training kernels' state (packed weights, a HIP graph), the captured sampling-step graphs with their static
        # buffers and the host mirrors of the schedule: caches of THIS module, not state — not copied
        for name in ('_dmh_train_step', '_graph_state', '_graph_states', '_host_cache'):
            ts = mod.__dict__.get(name)
            if ts is not None:
                memo[id(ts)] = None
    new = copy.deepcopy(model, memo)
    for mod in new.modules():
        for name in ('_dmh_train_step', '_graph_state', '_graph_states', '_host_cache'):
            mod.__dict__.pop(name, None)
    it = iter(engines)
    for mod in new.modules():
        if '_engine' in mod.__dict__:
            eng = next(it)
            mod._engine = type(eng)(mod, groups=eng.groups)
    return new


class EMA(nn.Module):
    """stand-in for ema_pytorch.EMA (package not installed here; its published algorithm restated): holds the
    sampling copy ``ema_model`` next to ``online_model`` with the same state_dict prefixes (``ema_model.`` /
    ``online_model.`` + ``initted``, ``step``).  Until the first ``update()`` the two share weights (a sampling-only
    run never pays for a second copy).  ``update()`` follows ema_pytorch: every ``update_every`` calls; copy while
    step <= update_after_step, afterwards ema <- lerp(ema, online, 1 - decay) with
    decay = clamp(1 - (1 + (step - update_after_step - 1) / inv_gamma) ** -power, min_value, beta)."""

    def __init__(self, model, beta=0.995, update_every=10, update_after_step=100, inv_gamma=1.0, power=2 / 3,
                 min_value=0.0):
        super().__init__()
        self.online_model = model
        self.ema_model = model            # sampling reads ema_model (DDP:1960); weights are shared until trained
        self.beta, self.update_every = beta, update_every
        self.update_after_step, self.inv_gamma, self.power, self.min_value = update_after_step, inv_gamma, power, min_value
        self.register_buffer('initted', torch.tensor(False))
        self.register_buffer('step', torch.tensor(0))

    def _own_copy(self):
        if self.ema_model is self.online_model:
            self.ema_model = _clone_without_engine_caches(self.online_model)
            self.ema_model.requires_grad_(False)

    def get_current_decay(self):
        epoch = max(int(self.step.item()) - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        return min(max(value, self.min_value), self.beta)

    def _pairs(self):
        on = dict(self.online_model.named_parameters())
        return [(p, on[k]) for k, p in self.ema_model.named_parameters()]

    def _bump(self):
        for mod in self.ema_model.modules():
            if '_engine' in mod.__dict__:
                mod._dmh_epoch = getattr(mod, '_dmh_epoch', 0) + 1

    def copy_params_from_model_to_ema(self):
        """ema_pytorch copies with ``.copy_()``: bit-exact, and a NaN / Inf left in the EMA copy does not survive it
        (a lerp with decay 0 would keep it)."""
        for pe, po in self._pairs():
            pe.data.copy_(po.data)
        self._bump()

    def update(self):
        from . import ops
        self._own_copy()
        step = int(self.step.item())
        self.step += 1
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not bool(self.initted.item()):
            self.copy_params_from_model_to_ema()
            self.initted.fill_(True)
        decay = self.get_current_decay()
        for pe, po in self._pairs():
            ops.ema_(pe.data, po.data, decay)
        self._bump()

    def state_dict(self, *a, **k):
        out = {'initted': self.initted, 'step': self.step}
        for key, v in self.online_model.state_dict().items():
            out['online_model.' + key] = v
        for key, v in self.ema_model.state_dict().items():
            out['ema_model.' + key] = v
        return out

    def load_state_dict(self, sd, strict=True):
        """accepts an ema_pytorch.EMA state_dict: ``ema_model.<key>``, optionally ``online_model.<key>`` (absent with
        include_online_model=False or in a stripped DGM.pt), ``initted``, ``step``.  Trainer.load has already put
        data['model'] into the online model; whether the EMA copy needs storage of its own is decided against THOSE
        weights, so an EMA-only checkpoint can never overwrite the online weights through the shared module."""
        ema = {k[len('ema_model.'):]: v for k, v in sd.items() if k.startswith('ema_model.')}
        online = {k[len('online_model.'):]: v for k, v in sd.items() if k.startswith('online_model.')}
        if self.ema_model is self.online_model and ema:
            cur = self.online_model.state_dict()
            if any(k in cur and (cur[k].shape != v.shape or not torch.equal(cur[k].detach().cpu(), v.detach().cpu()))
                   for k, v in ema.items()):
                self._own_copy()          # a checkpoint from a trained run: the two copies differ
        if 'step' in sd:
            self.step.fill_(int(sd['step']))
        if 'initted' in sd:
            self.initted.fill_(bool(sd['initted']))
        return self.ema_model.load_state_dict(ema, strict=strict)


class SyntheticConditions:
    """endless iterator of condition batches in the 12-channel layout of DDP:1162
    [img1(3) img2(3) mask(1) rgb_flow(3) flow(2)] built on device: seeded random
    640x360 homographies -> G1 -> K7 flow / HSV image; mask = union of 3 random rectangles; class 0."""

    def __init__(self, image_size, batch_size, seed=1000, device=None, first=0, stride=None):
        """sample i of the stream is built from ``seed + i``; this iterator yields samples first .. first + batch_size - 1,
        then advances by ``stride`` (default batch_size): rank r of N with first = r * bs, stride = N * bs walks the samples
        [b * bs * N + r * bs, + bs) of batch b — the ids ``distributed.noise_key`` gives the rank's noise, so a job's records
        do not depend on N."""
        self.image_size, self.batch_size, self.seed, self.count = image_size, batch_size, seed, int(first)
        self.stride = int(stride) if stride is not None else batch_size
        self.device = device

    def _homography(self, g):
        u = lambda a: (torch.rand((), generator=g, dtype=torch.float64).item() * 2 - 1) * a
        H0 = np.eye(3) + np.array([[u(.03), u(.03), u(8)], [u(.03), u(.03), u(8)], [u(3e-5), u(3e-5), 0.]])
        return adapt_homography_to_preprocessing_v3(360, 640, H0, self.image_size, self.image_size)

    def __iter__(self):
        return self

    def __next__(self):
        S, B = self.image_size, self.batch_size
        dev = self.device or _dev()
        homos, masks = [], torch.zeros((B, 1, S, S))
        for i in range(B):
            g = torch.Generator().manual_seed(self.seed + self.count + i)
            homos.append(self._homography(g))
            for _ in range(3):
                r = torch.rand(4, generator=g)
                y0, x0 = int(r[0] * S * 0.6), int(r[1] * S * 0.6)
                h, w = int((0.25 + 0.3 * r[2]) * S), int((0.25 + 0.3 * r[3]) * S)
                masks[i, 0, y0:y0 + h, x0:x0 + w] = 1.
        self.count += self.stride
        flow, rgb = homo_to_flow_rgb(np.stack(homos), S, S)
        data = torch.cat([torch.zeros((B, 6, S, S), device=dev), masks.to(dev), rgb, flow], dim=1)
        return data, torch.zeros((B,), dtype=torch.long, device=dev)


class Trainer(object):
    """DDP:1681-2021.  Keeps the constructor, ``save`` / ``load`` (checkpoint dict layout of DDP:1786-1802)
    and ``sample(idx, rank, step)``.  ``folder`` may be an iterator of (12-channel batch, classes) instead of the
    reference's dataset directory (the CA-Homo dataset path is outside the hot path); ``train`` is a §8f row."""

    def __init__(self, diffusion_model, folder, *, train_batch_size=16, gradient_accumulate_every=1,
                 augment_horizontal_flip=True, train_lr=1e-4, train_num_steps=100000, ema_update_every=10,
                 ema_decay=0.995, adam_betas=(0.9, 0.99), save_and_sample_every=1000, num_samples=9,
                 results_folder='./results', amp=False, fp16=False, split_batches=True, convert_image_to=None,
                 num_worker=8, total_data_slice_idx=1, data_slice_idx=1, shuffle=True, mixed_precision_type='fp16'):
        self.model = diffusion_model
        self.num_samples = num_samples
        self.save_and_sample_every = save_and_sample_every
        self.batch_size = train_batch_size
        self.gradient_accumulate_every = gradient_accumulate_every
        self.train_num_steps = train_num_steps
        self.image_size = diffusion_model.image_size
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)     # each rank reads its own slice
        # accelerate's Accelerator(split_batches=...) (DDP:1721-1722): True (the reference's default) = train_batch_size is
        # the GLOBAL batch, each process takes 1/world of every batch; False = every process takes train_batch_size
        self.split_batches = bool(split_batches)
        if self.split_batches and world > 1:
            if train_batch_size % world != 0:
                raise ValueError(f'train_batch_size ({train_batch_size}) must be a round multiple of the number of '
                                 f'processes ({world}) with split_batches=True (accelerate raises the same)')
            train_batch_size //= world
        self.rank_batch_size = train_batch_size
        if isinstance(folder, (str, os.PathLike)) and os.path.isfile(os.path.join(str(folder), 'BasesHomo_small.npy')):
            from .dataset import UnHomoTrainData, ConditionLoader                       # DDP:1735-1752
            self.ds = UnHomoTrainData(folder, self.image_size, augment_horizontal_flip=augment_horizontal_flip,
                                      convert_image_to=convert_image_to, workers=max(1, num_worker))
            self.dl = ConditionLoader(self.ds, train_batch_size, shuffle=shuffle, rank=rank, world=world)
        elif isinstance(folder, (str, os.PathLike)):
            # sample g of the JOB is built from seed 1000 + g whatever the number of ranks (rank r holds rows
            # [b * bs * world + r * bs, + bs) of batch b: the ids distributed.key_noise_by_sample keys its noise with)
            self.dl = SyntheticConditions(self.image_size, train_batch_size, seed=1000, first=rank * train_batch_size,
                                          stride=world * train_batch_size)
        else:
            self.dl = iter(folder)
        self.ema = EMA(diffusion_model, beta=ema_decay, update_every=ema_update_every)
        self.results_folder = Path(results_folder)
        self.step = 0
        self.train_lr, self.adam_betas = train_lr, adam_betas
        self._ts = None                   # train.TrainStep (Adam moments, packed training weights), built on first use
        self._opt_state = None            # an optimiser state_dict loaded before the first training step

    def save(self, milestone):
        self.results_folder.mkdir(exist_ok=True)
        data = {'step': self.step, 'model': self.model.state_dict(),
                'opt': self._ts.state_dict() if self._ts is not None else self._opt_state,
                'ema': self.ema.state_dict(), 'scaler': None, 'version': __version__}
        torch.save(data, str(self.results_folder / f'model-{milestone}.pt'))

    def load(self, milestone):
        data = torch.load(str(self.results_folder / f'model-{milestone}.pt'),
                          map_location=next(self.model.parameters()).device)
        self.model.load_state_dict(data['model'])
        self.step = data['step']
        if data.get('ema') is not None:
            self.ema.load_state_dict(data['ema'], strict=False)
        if data.get('opt') is not None:
            self._opt_state = data['opt']
            if self._ts is not None:
                self._ts.load_state_dict(self._opt_state)
        if self._ts is not None:
            self._ts.refresh()
        if 'version' in data:
            print(f"loading from version {data['version']}")

    def train_step_engine(self):
        if self._ts is None:
            from .train import TrainStep
            self._ts = TrainStep(self.model, lr=self.train_lr, betas=self.adam_betas,
                                 accum=self.gradient_accumulate_every)
            self.model.__dict__['_dmh_train_step'] = self._ts     # ``self.model(batch, classes=...)`` shares its packs
            if self._opt_state is not None:
                self._ts.load_state_dict(self._opt_state)
        return self._ts

    def train(self, log=None):
        """DDP:1828-1940: gradient accumulation, clip_grad_norm_(1.0), Adam, EMA, checkpoints (the latest as 9999 every
        500 steps, numbered ones every ``save_and_sample_every``).  The PNG/GIF sample dumps of DDP:1871-1935 are
        visualisation and stay out.  Ranks > 0 of a torch.distributed run train in lock-step (gradients averaged by
        one RCCL all-reduce per step) and leave EMA / checkpoints to rank 0, like accelerate's main process."""
        import torch.distributed as dist
        ts = self.train_step_engine()
        dev = next(self.model.parameters()).device
        main = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        while self.step < self.train_num_steps:
            batches = []
            for _ in range(self.gradient_accumulate_every):
                data = next(self.dl)
                batches.append((data[0].to(dev), data[1].to(dev)))
            total_loss = ts.step(batches)
            self.step += 1
            if log is not None:
                log(self.step, total_loss)
            if main:
                self.ema.update()
                if self.step % 500 == 0:
                    self.save(9999)
                if self.step % self.save_and_sample_every == 0:
                    self.save(self.step // self.save_and_sample_every)
        if main:
            print('training complete')

    def sample(self, idx, rank, step=1):
        """DDP:1941-2021 without the every-100-steps PNG/GIF dumps (visualisation is out of scope)."""
        data = next(self.dl)
        dev = torch.device('cuda', rank) if isinstance(rank, int) else torch.device(rank)
        rgb_flows = data[0][:, -5:-2].to(dev)
        flows = data[0][:, -2:].to(dev).contiguous()
        mask = data[0][:, -6:-5].to(dev)
        with torch.no_grad():
            all_images = self.ema.ema_model.sample(classes=data[1].to(dev), rgb_flow=rgb_flows, flow=flows, mask=mask)
        return saveTrainPair(all_images[0], mask=all_images[1], flows=all_images[2])

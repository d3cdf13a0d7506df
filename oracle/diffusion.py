"""Oracle: diffusion process (test infrastructure, CPU, plain PyTorch).

Restates GaussianDiffusion of the reference:
  CFG = classifier_free_guidance.py:472-842   DDP = denoising_diffusion_pytorch.py:453-817
All RNG draws go through an ``rng`` object so tests can record / replay the
exact stream (randn(shape), then per step uniform_(B) and,
except on the last step, randn_like).
"""
import math

import torch

from . import unet as U


# ------------------------------------------------------------------ RNG plumbing
class TorchRng:
    """draws from torch's global CPU generator in the reference's call order."""

    def randn(self, shape):
        return torch.randn(tuple(shape))

    def uniform(self, n):
        return torch.zeros((n,)).float().uniform_(0, 1)      # CFG:90


class ReplayRng:
    """replays a recorded list of tensors (order = draw order)."""

    def __init__(self, draws):
        self.draws = [torch.as_tensor(d) for d in draws]
        self.i = 0

    def _next(self):
        d = self.draws[self.i]
        self.i += 1
        return d

    def randn(self, shape):
        d = self._next()
        assert tuple(d.shape) == tuple(shape), (d.shape, shape)
        return d

    def uniform(self, n):
        d = self._next()
        assert d.shape == (n,)
        return d


class RecordRng(TorchRng):
    def __init__(self):
        self.draws = []

    def randn(self, shape):
        d = super().randn(shape)
        self.draws.append(d.clone())
        return d

    def uniform(self, n):
        d = super().uniform(n)
        self.draws.append(d.clone())
        return d


# ------------------------------------------------------------------ D1
def linear_beta_schedule(timesteps):
    """CFG:478-482 (float64)."""
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps, s=0.008):
    """CFG:485-495 (float64)."""
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


# ------------------------------------------------------------------ D2
BUFFER_NAMES = (
    'betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
    'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod',
    'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'posterior_variance',
    'posterior_log_variance_clipped', 'posterior_mean_coef1', 'posterior_mean_coef2',
    'p2_loss_weight')


def schedule_buffers(timesteps=1000, beta_schedule='cosine', p2_loss_weight_gamma=0.,
                     p2_loss_weight_k=1):
    """the 13 registered buffers of GaussianDiffusion.__init__, CFG:528-584:
    computed in float64, stored as float32."""
    if beta_schedule == 'linear':
        betas = linear_beta_schedule(timesteps)
    elif beta_schedule == 'cosine':
        betas = cosine_beta_schedule(timesteps)
    else:
        raise ValueError(f'unknown beta schedule {beta_schedule}')
    alphas = 1. - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.cat((torch.ones(1, dtype=torch.float64), ac[:-1]))
    post_var = betas * (1. - ac_prev) / (1. - ac)
    b64 = {
        'betas': betas,
        'alphas_cumprod': ac,
        'alphas_cumprod_prev': ac_prev,
        'sqrt_alphas_cumprod': torch.sqrt(ac),
        'sqrt_one_minus_alphas_cumprod': torch.sqrt(1. - ac),
        'log_one_minus_alphas_cumprod': torch.log(1. - ac),
        'sqrt_recip_alphas_cumprod': torch.sqrt(1. / ac),
        'sqrt_recipm1_alphas_cumprod': torch.sqrt(1. / ac - 1),
        'posterior_variance': post_var,
        'posterior_log_variance_clipped': torch.log(post_var.clamp(min=1e-20)),
        'posterior_mean_coef1': betas * torch.sqrt(ac_prev) / (1. - ac),
        'posterior_mean_coef2': (1. - ac_prev) * torch.sqrt(alphas) / (1. - ac),
        'p2_loss_weight': (p2_loss_weight_k + ac / (1 - ac)) ** -p2_loss_weight_gamma,
    }
    return {k: v.to(torch.float32) for k, v in b64.items()}


# ------------------------------------------------------------------ D3/D4
def extract(a, t, ndim):
    """CFG:472-475."""
    return a.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def ddim_time_pairs(num_timesteps, sampling_timesteps):
    """CFG:674-677: linspace(-1, T-1, S+1) -> int (truncation) -> reversed pairs."""
    times = torch.linspace(-1, num_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def _predictions(buf, objective, model_output, x, t, clip):
    """model_predictions body after the network call, CFG:612-630."""
    nd = x.dim()
    rc = extract(buf['sqrt_recip_alphas_cumprod'], t, nd)
    rm1 = extract(buf['sqrt_recipm1_alphas_cumprod'], t, nd)
    clamp = (lambda v: torch.clamp(v, min=-1., max=1.)) if clip else (lambda v: v)
    if objective == 'pred_noise':
        pred_noise = model_output
        x_start = clamp(rc * x - rm1 * pred_noise)                       # CFG:586-588
    elif objective == 'pred_x0':
        x_start = clamp(model_output)
        pred_noise = (rc * x - x_start) / rm1                            # CFG:590-594
    elif objective == 'pred_v':
        sa = extract(buf['sqrt_alphas_cumprod'], t, nd)
        s1 = extract(buf['sqrt_one_minus_alphas_cumprod'], t, nd)
        x_start = clamp(sa * x - s1 * model_output)                      # CFG:600-601
        pred_noise = (rc * x - x_start) / rm1
    else:
        raise ValueError(objective)
    return pred_noise, x_start


def _ddim_update(buf, x_start, pred_noise, time, time_next, eta, noise):
    """CFG:697-707 — 0-dim fp32 tensor arithmetic, same op order."""
    alpha = buf['alphas_cumprod'][time]
    alpha_next = buf['alphas_cumprod'][time_next]
    sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
    c = (1 - alpha_next - sigma ** 2).sqrt()
    return x_start * alpha_next.sqrt() + c * pred_noise + sigma * noise


def ddim_coefficients(buf, time, time_next, eta=1.):
    """the three fp32 scalars of one DDIM update (sqrt(alpha_next), c, sigma)."""
    alpha = buf['alphas_cumprod'][time]
    alpha_next = buf['alphas_cumprod'][time_next]
    sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
    c = (1 - alpha_next - sigma ** 2).sqrt()
    return float(alpha_next.sqrt()), float(c), float(sigma)


# ------------------------------------------------------------------ D6/D8 (CFG)
def cfg_ddim_sample(sd, buf, classes, rgb_flow, flow, mask, shape, *, sampling_timesteps,
                    objective='pred_x0', cond_scale=3., cond_drop_prob=0.5, eta=1.,
                    clip_denoised=True, groups=8, rng=None, trace=None):
    """CFG GaussianDiffusion.ddim_sample, CFG:669-711.  ``sd`` = Unet state_dict.

    The conditional pass draws ``uniform(B) < 1 - cond_drop_prob`` each step
    (CFG:415,421-422,90) when 0 < cond_drop_prob < 1."""
    rng = rng or TorchRng()
    T = buf['betas'].shape[0]
    b = shape[0]
    img = rng.randn(shape)
    for time, time_next in ddim_time_pairs(T, sampling_timesteps):
        t = torch.full((b,), time, dtype=torch.long)
        if cond_drop_prob > 0:
            p_keep = 1 - cond_drop_prob
            if p_keep == 1:
                keep = torch.ones(b, dtype=torch.bool)
            elif p_keep == 0:
                keep = torch.zeros(b, dtype=torch.bool)
            else:
                keep = rng.uniform(b) < p_keep
        else:
            keep = None
        out = U.cfg_unet_forward_with_cond_scale(sd, img, t, classes, rgb_flow, mask, keep,
                                                 cond_scale, groups)
        pred_noise, x_start = _predictions(buf, objective, out, img, t, clip_denoised)
        if trace is not None:
            trace.append({'time': time, 'keep': keep, 'model_out': out, 'x_start': x_start})
        if time_next < 0:
            img = x_start
        else:
            img = _ddim_update(buf, x_start, pred_noise, time, time_next, eta, rng.randn(shape))
        if trace is not None:
            trace[-1]['img'] = img
    return (img + 1) * 0.5, mask, flow


def cfg_sample(sd, buf, classes, rgb_flow, flow, mask, *, image_size, channels, **kw):
    """CFG GaussianDiffusion.sample, CFG:713-720: rgb_flow -> [-1,1], DDIM only
    (the p_sample_loop branch of CFG:719 cannot be called — SURVEY.md fact 6)."""
    rgb_flow = rgb_flow * 2 - 1
    shape = (classes.shape[0], channels, image_size, image_size)
    return cfg_ddim_sample(sd, buf, classes, rgb_flow, flow, mask, shape, **kw)


# ------------------------------------------------------------------ D6/D7/D8 (DDP)
def _ddp_model(sd, x, t, self_cond, self_condition, groups):
    return U.ddp_unet_forward(sd, x, t, self_cond, self_condition, groups)


def ddp_p_sample(sd, buf, x, t, *, objective='pred_noise', self_cond=None, self_condition=False,
                 clip_denoised=True, groups=8, rng=None):
    """DDP p_sample / p_mean_variance / q_posterior, DDP:636-661,604-611."""
    rng = rng or TorchRng()
    bt = torch.full((x.shape[0],), t, dtype=torch.long)
    out = _ddp_model(sd, x, bt, self_cond, self_condition, groups)
    _, x_start = _predictions(buf, objective, out, x, bt, False)         # DDP:637 (no clip arg)
    if clip_denoised:
        x_start = x_start.clamp(-1., 1.)                                 # DDP:640-641
    nd = x.dim()
    mean = (extract(buf['posterior_mean_coef1'], bt, nd) * x_start +
            extract(buf['posterior_mean_coef2'], bt, nd) * x)
    logvar = extract(buf['posterior_log_variance_clipped'], bt, nd)
    noise = rng.randn(x.shape) if t > 0 else 0.
    return mean + (0.5 * logvar).exp() * noise, x_start


def ddp_p_sample_loop(sd, buf, shape, *, self_condition=False, trace=None, **kw):
    """DDP p_sample_loop, DDP:663-680 (last two channels remapped x*2-1)."""
    rng = kw.pop('rng', None) or TorchRng()
    T = buf['betas'].shape[0]
    img = rng.randn(shape)
    x_start = None
    for t in reversed(range(T)):
        sc = x_start if self_condition else None
        img, x_start = ddp_p_sample(sd, buf, img, t, self_cond=sc, self_condition=self_condition,
                                    rng=rng, **kw)
        if trace is not None:
            trace.append({'t': t, 'img': img, 'x_start': x_start})
    img = (img + 1) * 0.5
    img = img.clone()
    img[:, -2:] = img[:, -2:] * 2 - 1
    return img


def ddp_ddim_sample(sd, buf, shape, *, sampling_timesteps, objective='pred_noise', eta=1.,
                    self_condition=False, clip_denoised=True, groups=8, rng=None, trace=None):
    """DDP ddim_sample, DDP:682-729 (last two channels remapped (x*2-1)*512)."""
    rng = rng or TorchRng()
    T = buf['betas'].shape[0]
    b = shape[0]
    img = rng.randn(shape)
    x_start = None
    for time, time_next in ddim_time_pairs(T, sampling_timesteps):
        t = torch.full((b,), time, dtype=torch.long)
        sc = x_start if self_condition else None
        out = _ddp_model(sd, img, t, sc, self_condition, groups)
        pred_noise, x_start = _predictions(buf, objective, out, img, t, clip_denoised)
        if time_next < 0:
            img = x_start
        else:
            img = _ddim_update(buf, x_start, pred_noise, time, time_next, eta, rng.randn(shape))
        if trace is not None:
            trace.append({'time': time, 'img': img, 'x_start': x_start})
    img = (img + 1) * 0.5
    img = img.clone()
    img[:, -2:] = (img[:, -2:] * 2 - 1) * 512
    return img


def ddp_interpolate(sd, buf, x1, x2, t=None, lam=0.5, *, rng=None, **kw):
    """D10, DDP:737-754: q_sample both ends at step t (two randn_like draws, x1's first), blend
    ``(1 - lam) * xt1 + lam * xt2`` in that op order, then p_sample down from t-1 to 0.
    The reference's loop (DDP:748-752) hands p_sample a (b,) TENSOR where it takes ``t: int`` (DDP:648: ``torch.full``
    rejects it) and assigns p_sample's (pred_img, x_start) tuple back to ``img``; it therefore raises for every t > 0.
    The evident intent — python-int steps, carry pred_img — is restated here; for t = 0 (no loop) this is the
    reference's own result (tests/golden/interpolate.npz holds both: `t0.*` from the reference's interpolate as it
    stands, `t3.*` from the reference's q_sample / p_sample called in this chain)."""
    rng = rng or TorchRng()
    T = buf['betas'].shape[0]
    t = T - 1 if t is None else t
    assert x1.shape == x2.shape
    tb = torch.full((x1.shape[0],), t, dtype=torch.long)
    xt1 = q_sample(buf, x1, tb, rng.randn(x1.shape))
    xt2 = q_sample(buf, x2, tb, rng.randn(x2.shape))
    img = (1 - lam) * xt1 + lam * xt2
    for i in reversed(range(0, t)):
        img, _ = ddp_p_sample(sd, buf, img, i, rng=rng, **kw)
    return img


# ------------------------------------------------------------------ D9
def q_sample(buf, x_start, t, noise):
    """CFG:738-742."""
    nd = x_start.dim()
    return (extract(buf['sqrt_alphas_cumprod'], t, nd) * x_start +
            extract(buf['sqrt_one_minus_alphas_cumprod'], t, nd) * noise)


def ddp_p_losses(sd, buf, x_start, t, noise, *, objective='pred_noise', loss_type='l1', self_condition=False,
                 use_self_cond=False, groups=8):
    """DDP GaussianDiffusion.p_losses, DDP:772-811 (forward value).  ``use_self_cond`` is the outcome of the reference's
    ``random() < 0.5`` draw (DDP:785): when set (and the model is self-conditioned) the network first predicts x_start
    without a self-condition (model_predictions, DDP:787, no clamp) and is then run again conditioned on it."""
    import torch.nn.functional as F
    x = q_sample(buf, x_start, t, noise)
    x_self_cond = None
    if self_condition and use_self_cond:
        out0 = _ddp_model(sd, x, t, None, self_condition, groups)
        _, x_self_cond = _predictions(buf, objective, out0, x, t, False)
    model_out = _ddp_model(sd, x, t, x_self_cond, self_condition, groups)
    if objective == 'pred_noise':
        target = noise
    elif objective == 'pred_x0':
        target = x_start
    elif objective == 'pred_v':
        nd = x_start.dim()
        target = (extract(buf['sqrt_alphas_cumprod'], t, nd) * noise -
                  extract(buf['sqrt_one_minus_alphas_cumprod'], t, nd) * x_start)       # predict_v, DDP:596-598
    else:
        raise ValueError(f'unknown objective {objective}')
    loss = (F.l1_loss if loss_type == 'l1' else F.mse_loss)(model_out, target, reduction='none')
    # einops' reduce(loss, 'b ... -> b (...)', 'mean') reduces nothing (it only flattens, DDP:808): every element is
    # weighted by its sample's p2 weight and ONE mean runs over all of them (DDP:810-811)
    loss = loss.reshape(loss.shape[0], -1)
    loss = loss * extract(buf['p2_loss_weight'], t, loss.dim())
    return loss.mean()


def cfg_p_losses(sd, buf, x_start, t, classes, rgb_flow, flow, mask, noise, keep_mask, *, objective='pred_x0',
                 loss_type='l1', groups=8):
    """CFG GaussianDiffusion.p_losses, CFG:770-806 (forward value only).  ``keep_mask`` is the class-dropout
    draw of the network call (CFG:780 -> CFG:415,422).  einops' ``reduce(x, 'b ... -> b (...)', 'mean')`` keeps
    every axis (it only flattens), so both terms end in a plain mean over all elements."""
    from . import geometry as G
    import torch.nn.functional as F
    x = q_sample(buf, x_start, t, noise)
    out = U.cfg_unet_forward(sd, x, t, classes, rgb_flow, mask, keep_mask, groups)
    im1, im2 = out[:, :3], out[:, 3:]
    im2_warp = G.flow_warp(im2.contiguous(), flow)
    if objective == 'pred_noise':
        target = noise
    elif objective == 'pred_x0':
        target = x_start
    elif objective == 'pred_v':
        target = (extract(buf['sqrt_alphas_cumprod'], t, 4) * noise -
                  extract(buf['sqrt_one_minus_alphas_cumprod'], t, 4) * x_start)           # CFG:596-598
    else:
        raise ValueError(f'unknown objective {objective}')
    fn = {'l1': F.l1_loss, 'l2': F.mse_loss}[loss_type]
    loss = fn(out, target, reduction='none').reshape(out.shape[0], -1)
    photo = (mask * fn(im2_warp, im1, reduction='none')).reshape(out.shape[0], -1)
    w = extract(buf['alphas_cumprod'], t, 2)
    return loss.mean() + (1 * w * photo).mean()


def cfg_forward_split(img):
    """CFG GaussianDiffusion.forward's slicing of the 12-channel batch, CFG:814-840 (DDP:1162 layout)."""
    data, mask, rgb_flow, flow = img[:, :6], img[:, 6:7], img[:, -5:-2], img[:, -2:]
    return data * 2 - 1, mask, rgb_flow * 2 - 1, flow

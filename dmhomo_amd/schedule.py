"""Noise schedules and the 13 schedule buffers (host side, float64 -> float32).

D1/D2 of SURVEY.md §8a: CFG:478-495,528-584 / DDP:460-478,513-582.  These run once
at construction on the host (as in the reference); their float32 images are part
of ``GaussianDiffusion.state_dict()`` and feed the sampler kernels as scalars.
"""
import math

import torch

BUFFER_ORDER = (
    'betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
    'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod',
    'sqrt_recipm1_alphas_cumprod', 'posterior_variance', 'posterior_log_variance_clipped',
    'posterior_mean_coef1', 'posterior_mean_coef2', 'p2_loss_weight')


def linear_beta_schedule(timesteps):
    """CFG:478-482 / DDP:460-464 (float64, host)."""
    k = 1000 / timesteps
    return torch.linspace(k * 1e-4, k * 2e-2, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps, s=0.008):
    """CFG:485-495 / DDP:467-478 (float64, host): abar = cos^2(((x / T) + s) / (1 + s) * pi / 2), normalised at 0."""
    grid = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    abar = torch.cos((grid / timesteps + s) / (1 + s) * math.pi * 0.5) ** 2
    abar = abar / abar[0]
    return torch.clip(1 - abar[1:] / abar[:-1], 0, 0.999)


def betas_for(name, timesteps):
    if name == 'linear':
        return linear_beta_schedule(timesteps)
    if name == 'cosine':
        return cosine_beta_schedule(timesteps)
    raise ValueError(f'unknown beta schedule {name}')


def make_buffers(beta_schedule, timesteps, p2_gamma=0., p2_k=1):
    """ordered dict name -> float32 tensor (T,) in the reference's registration order."""
    betas = betas_for(beta_schedule, timesteps)
    alphas = 1. - betas
    abar = torch.cumprod(alphas, dim=0)
    abar_prev = torch.nn.functional.pad(abar[:-1], (1, 0), value=1.)
    one_m = 1. - abar
    pvar = betas * (1. - abar_prev) / one_m
    vals = (betas, abar, abar_prev, torch.sqrt(abar), torch.sqrt(one_m), torch.log(one_m), torch.sqrt(1. / abar),
            torch.sqrt(1. / abar - 1), pvar, torch.log(pvar.clamp(min=1e-20)),
            betas * torch.sqrt(abar_prev) / one_m, (1. - abar_prev) * torch.sqrt(alphas) / one_m,
            (p2_k + abar / (1 - abar)) ** -p2_gamma)
    return {n: v.to(torch.float32) for n, v in zip(BUFFER_ORDER, vals)}


def ddim_pairs(num_timesteps, sampling_timesteps):
    """(t, t_next) pairs of the DDIM loop, last t_next == -1 (CFG:674-677)."""
    ts = torch.linspace(-1, num_timesteps - 1, steps=sampling_timesteps + 1).int().tolist()
    ts.reverse()
    return list(zip(ts[:-1], ts[1:]))

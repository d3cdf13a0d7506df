"""Deterministic, construction-order-independent weights for parity tests.

Every tensor of a ``state_dict`` is regenerated from its *key name* and shape
with a CPU ``torch.Generator`` seeded by crc32(key) ^ seed, so the golden
generator (run where /root/reference exists) and the tests (run anywhere) agree
on the weights without shipping them.  Norm gains are 1 + 0.2 N(0,1), biases
0.1 N(0,1), embeddings N(0,1), conv / linear weights N(0,1)/sqrt(fan_in).
A float64 checksum of the generated tensors is stored in each fixture.
"""
import zlib

import torch

SCHEDULE_KEYS = (
    'betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
    'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod',
    'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod', 'posterior_variance',
    'posterior_log_variance_clipped', 'posterior_mean_coef1', 'posterior_mean_coef2',
    'p2_loss_weight')


def det_tensor(key, shape, seed=0):
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7fffffff)
    r = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    leaf = key.split('.')[-1]
    if leaf == 'g' or key.endswith('norm.weight'):
        return 1 + 0.2 * r
    if leaf == 'bias':
        return 0.1 * r
    if 'classes_emb' in key:
        return r
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return r / max(fan_in, 1) ** 0.5


def det_state_dict(shapes, seed=0, out_scale=1.0):
    """shapes: {key: shape} (e.g. from a module's state_dict). Schedule buffers are skipped.
    out_scale: factor on ``final_conv.{weight,bias}``.  With 1.0 the dim-64 UNet's output reaches |8..10| and 62-73 % of a
    pred_x0 sampler's x_start is clamped to +-1 (CFG:612) — a clamped element compares equal whatever the kernels computed.
    0.15 keeps the output inside ~[-1.5, 1.5] (< 5 % clamped, measured ~0.5 %): the DE-SATURATED weight set of the
    end-to-end sampler parity tests."""
    sd = {k: det_tensor(k, s, seed) for k, s in sorted(shapes.items())
          if k.split('.')[-1] not in SCHEDULE_KEYS}
    if out_scale != 1.0:
        for k in sd:
            if k.endswith('final_conv.weight') or k.endswith('final_conv.bias'):
                sd[k] = sd[k] * out_scale
    return sd


DESAT = 0.15     # out_scale of the de-saturated weight set


def checksum(sd):
    return float(sum(v.double().abs().sum() for _, v in sorted(sd.items())))


def shapes_of(module_or_sd):
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, 'state_dict') else module_or_sd
    return {k: tuple(v.shape) for k, v in sd.items()}

"""Parameter containers with the reference's ``state_dict`` layout.

The modules built here only *hold* parameters (stock ``torch.nn`` containers are
used as holders so default initialisation and key names match the reference:
SURVEY.md §8b checkpoint contract).  None of their ``forward`` methods is ever
called — arithmetic happens in ``dmhomo_amd.engine`` on the HIP kernels.
Construction order follows the reference (CFG:333-401, DDP:338-406) so a given
``torch.manual_seed`` yields the same initial weights.
"""
from torch import nn
import torch

HIDDEN = 128      # heads * dim_head = 4 * 32 (CFG:246-250)


class Holder(nn.Module):
    """a parameter namespace; not callable."""

    def forward(self, *a, **k):
        raise RuntimeError('dmhomo_amd parameter holders are not executable modules')


def time_embedding(dim, learned_sinusoidal_cond, random_fourier_features, learned_sinusoidal_dim):
    """slot 0 of ``time_mlp`` and the width it feeds time_mlp.1 with (CFG:344-353 / DDP:348-364): SinusoidalPosEmb has no
    parameters; RandomOrLearnedSinusoidalPosEmb holds ``weights`` (half_dim,), frozen when random (CFG:182-183)."""
    h = Holder()
    if learned_sinusoidal_cond or random_fourier_features:
        assert learned_sinusoidal_dim % 2 == 0
        h.weights = nn.Parameter(torch.randn(learned_sinusoidal_dim // 2), requires_grad=not random_fourier_features)
        return h, learned_sinusoidal_dim + 1
    return h, dim


def gain(dim):
    h = Holder()
    h.g = nn.Parameter(torch.ones(1, dim, 1, 1))             # LayerNorm.g, CFG:135
    return h


def block(dim, dim_out, groups):
    h = Holder()
    h.proj = nn.Conv2d(dim, dim_out, 3, padding=1)            # WeightStandardizedConv2d params, CFG:200
    h.norm = nn.GroupNorm(groups, dim_out)
    return h


def resnet_block(dim, dim_out, emb_dim, groups):
    h = Holder()
    h.mlp = nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, dim_out * 2))     # CFG:220 / DDP:225
    h.block1 = block(dim, dim_out, groups)
    h.block2 = block(dim_out, dim_out, groups)
    if dim != dim_out:
        h.res_conv = nn.Conv2d(dim, dim_out, 1)
    return h


def _prenorm_residual(dim, inner):
    outer, pre = Holder(), Holder()
    pre.fn = inner
    pre.norm = gain(dim)
    outer.fn = pre
    return outer


def linear_attention(dim):
    a = Holder()
    a.to_qkv = nn.Conv2d(dim, HIDDEN * 3, 1, bias=False)
    a.to_out = nn.Sequential(nn.Conv2d(HIDDEN, dim, 1), gain(dim))        # CFG:251-253
    return _prenorm_residual(dim, a)


def attention(dim):
    a = Holder()
    a.to_qkv = nn.Conv2d(dim, HIDDEN * 3, 1, bias=False)
    a.to_out = nn.Conv2d(HIDDEN, dim, 1)                                   # CFG:281-282
    return _prenorm_residual(dim, a)


def build_trunk(m, dim, init_dim, dim_mults, in_channels, emb_dim, groups, out_dim, downsample):
    """attach init_conv .. final_conv to module ``m`` (after its embedding MLPs)."""
    dims = [init_dim, *[dim * mult for mult in dim_mults]]
    in_out = list(zip(dims[:-1], dims[1:]))
    n = len(in_out)
    m.downs = nn.ModuleList([])
    m.ups = nn.ModuleList([])
    for ind, (d_in, d_out) in enumerate(in_out):
        last = ind >= n - 1
        m.downs.append(nn.ModuleList([
            resnet_block(d_in, d_in, emb_dim, groups),
            resnet_block(d_in, d_in, emb_dim, groups),
            linear_attention(d_in),
            downsample(d_in, d_out) if not last else nn.Conv2d(d_in, d_out, 3, padding=1)]))
    mid = dims[-1]
    m.mid_block1 = resnet_block(mid, mid, emb_dim, groups)
    m.mid_attn = attention(mid)
    m.mid_block2 = resnet_block(mid, mid, emb_dim, groups)
    for ind, (d_in, d_out) in enumerate(reversed(in_out)):
        last = ind == n - 1
        m.ups.append(nn.ModuleList([
            resnet_block(d_out + d_in, d_out, emb_dim, groups),
            resnet_block(d_out + d_in, d_out, emb_dim, groups),
            linear_attention(d_out),
            nn.Sequential(Holder(), nn.Conv2d(d_out, d_in, 3, padding=1)) if not last
            else nn.Conv2d(d_out, d_in, 3, padding=1)]))
    m.final_res_block = resnet_block(dim * 2, dim, emb_dim, groups)
    m.final_conv = nn.Conv2d(dim, out_dim, 1)


def downsample_cfg(d_in, d_out):
    return nn.Conv2d(d_in, d_out, 4, 2, 1)                                 # CFG:110-111


def downsample_ddp(d_in, d_out):
    return nn.Sequential(Holder(), nn.Conv2d(d_in * 4, d_out, 1))          # DDP:110-113

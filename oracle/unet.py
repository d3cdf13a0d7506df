"""Oracle: UNet forward passes (test infrastructure, CPU fp32, plain PyTorch).

Functional restatement over a ``state_dict`` of the two UNets of the reference:

* CFG = DGM/denoising_diffusion_models/classifier_free_guidance.py
        (conditional: class embedding + ``rgb_flow * mask`` concat; CFG:302-466)
* DDP = DGM/denoising_diffusion_models/denoising_diffusion_pytorch.py
        (unconditional, optional self-conditioning; DDP:315-447)

Every function cites the reference lines whose arithmetic it follows.  The op
order (which ATen op is called on which operand) is kept identical so that on
the same host the oracle is bit-equal to the reference; tests pin that against
``tests/golden/*.npz``.
"""
import math

import torch
import torch.nn.functional as F

HEADS = 4          # CFG:246,275 (never overridden by the reference's Unet)
DIM_HEAD = 32


def _sub(sd, prefix):
    """view of a state_dict under ``prefix.``"""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix + '.')}


# --------------------------------------------------------------------------- N1
def ws_conv3x3(x, w, b):
    """WeightStandardizedConv2d.forward, CFG:120-128 / DDP:122-132.

    per-out-channel biased variance over (Cin,kh,kw); eps 1e-5 for fp32."""
    eps = 1e-5 if x.dtype == torch.float32 else 1e-3
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = torch.var(w, dim=(1, 2, 3), unbiased=False, keepdim=True)
    wn = (w - mean) * (var + eps).rsqrt()
    return F.conv2d(x, wn, b, 1, 1)


def ws_fold(w, eps=1e-5):
    """the standardized weight alone (what the HIP path pre-folds at load)."""
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = torch.var(w, dim=(1, 2, 3), unbiased=False, keepdim=True)
    return (w - mean) * (var + eps).rsqrt()


# --------------------------------------------------------------------------- N2
def block(p, x, groups, scale_shift=None):
    """Block.forward, CFG:204-213: WS conv -> GroupNorm -> (scale+1, shift) -> SiLU."""
    y = ws_conv3x3(x, p['proj.weight'], p['proj.bias'])
    y = F.group_norm(y, groups, p['norm.weight'], p['norm.bias'], 1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        y = y * (scale + 1) + shift
    return F.silu(y)


# --------------------------------------------------------------------------- N3
def resnet_block(p, x, groups, cond_emb=None):
    """ResnetBlock.forward, CFG:227-241 / DDP:233-245.

    ``cond_emb`` is cat(time_emb, class_emb) for CFG (CFG:231-232) or the time
    embedding alone for DDP (DDP:236-237)."""
    ss = None
    if cond_emb is not None and 'mlp.1.weight' in p:
        e = F.linear(F.silu(cond_emb), p['mlp.1.weight'], p['mlp.1.bias'])
        e = e[:, :, None, None]
        ss = e.chunk(2, dim=1)
    h = block(_sub(p, 'block1'), x, groups, ss)
    h = block(_sub(p, 'block2'), h, groups)
    if 'res_conv.weight' in p:
        res = F.conv2d(x, p['res_conv.weight'], p['res_conv.bias'])
    else:
        res = x
    return h + res


# --------------------------------------------------------------------------- N4
def chan_layernorm(x, g):
    """LayerNorm.forward over the channel dim, CFG:137-141 (biased var, gain only)."""
    eps = 1e-5 if x.dtype == torch.float32 else 1e-3
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + eps).rsqrt() * g


# --------------------------------------------------------------------------- N5
def linear_attention(p, x):
    """LinearAttention.forward, CFG:255-270."""
    b, c, h, w = x.shape
    n = h * w
    qkv = F.conv2d(x, p['to_qkv.weight']).chunk(3, dim=1)
    q, k, v = (t.reshape(b, HEADS, DIM_HEAD, n) for t in qkv)
    q = q.softmax(dim=-2)
    k = k.softmax(dim=-1)
    q = q * (DIM_HEAD ** -0.5)
    v = v / n
    ctx = torch.einsum('b h d n, b h e n -> b h d e', k, v)
    out = torch.einsum('b h d e, b h d n -> b h e n', ctx, q)
    out = out.reshape(b, HEADS * DIM_HEAD, h, w)
    out = F.conv2d(out, p['to_out.0.weight'], p['to_out.0.bias'])
    return chan_layernorm(out, p['to_out.1.g'])


# --------------------------------------------------------------------------- N6
def attention(p, x):
    """Attention.forward, CFG:284-296 (scale on q before QK^T, softmax over keys)."""
    b, c, h, w = x.shape
    n = h * w
    qkv = F.conv2d(x, p['to_qkv.weight']).chunk(3, dim=1)
    q, k, v = (t.reshape(b, HEADS, DIM_HEAD, n) for t in qkv)
    q = q * (DIM_HEAD ** -0.5)
    sim = torch.einsum('b h d i, b h d j -> b h i j', q, k)
    attn = sim.softmax(dim=-1)
    out = torch.einsum('b h i j, b h d j -> b h i d', attn, v)
    out = out.permute(0, 1, 3, 2).reshape(b, HEADS * DIM_HEAD, h, w)
    return F.conv2d(out, p['to_out.weight'], p['to_out.bias'])


def _res_prenorm(p, x, fn):
    """Residual(PreNorm(dim, fn)), CFG:96-103,144-153."""
    return fn(_sub(p, 'fn.fn'), chan_layernorm(x, p['fn.norm.g'])) + x


# --------------------------------------------------------------------------- N7/N8
def sinusoidal_pos_emb(t, dim):
    """SinusoidalPosEmb.forward, CFG:165-172; ``t`` int64 (B,)."""
    half = dim // 2
    f = math.log(10000) / (half - 1)
    f = torch.exp(torch.arange(half, device=t.device) * -f)
    e = t[:, None] * f[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def random_or_learned_pos_emb(t, weights):
    """RandomOrLearnedSinusoidalPosEmb.forward, CFG:185-190 [DDP:190-195]: cat(t, sin(t w 2 pi), cos(t w 2 pi)); ``t`` int64
    (B,), ``weights`` (half,) — the op order of the reference (x * w, * 2, * math.pi) is kept."""
    x = t[:, None]
    freqs = x * weights[None, :] * 2 * math.pi
    return torch.cat((x, torch.cat((freqs.sin(), freqs.cos()), dim=-1)), dim=-1)


def time_mlp(sd, t, dim):
    """time_mlp = SinusoidalPosEmb (or, with ``time_mlp.0.weights`` in the state_dict, RandomOrLearnedSinusoidalPosEmb:
    CFG:346-351) -> Linear -> GELU(erf) -> Linear, CFG:353."""
    if 'time_mlp.0.weights' in sd:
        e = random_or_learned_pos_emb(t, sd['time_mlp.0.weights'])
    else:
        e = sinusoidal_pos_emb(t, dim)
    e = F.linear(e, sd['time_mlp.1.weight'], sd['time_mlp.1.bias'])
    e = F.gelu(e)
    return F.linear(e, sd['time_mlp.3.weight'], sd['time_mlp.3.bias'])


def class_mlp(sd, classes, keep_mask):
    """classes_emb / null_classes_emb / classes_mlp, CFG:419-427.

    ``keep_mask`` (B,) bool or None; None means cond_drop_prob == 0 (no swap)."""
    e = F.embedding(classes, sd['classes_emb.weight'])
    if keep_mask is not None:
        null = sd['null_classes_emb'][None, :].expand(e.shape[0], -1)
        e = torch.where(keep_mask[:, None], e, null)
    e = F.linear(e, sd['classes_mlp.0.weight'], sd['classes_mlp.0.bias'])
    e = F.gelu(e)
    return F.linear(e, sd['classes_mlp.2.weight'], sd['classes_mlp.2.bias'])


# --------------------------------------------------------------------------- N9
def _downsample(p, x):
    """CFG:110-111 4x4/s2/p1 conv, DDP:110-113 pixel-unshuffle + 1x1, or the
    last stage's plain 3x3 conv (CFG:378)."""
    if '1.weight' in p:                              # DDP Sequential(Rearrange, Conv2d)
        b, c, h, w = x.shape
        y = x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4)
        y = y.reshape(b, c * 4, h // 2, w // 2)
        return F.conv2d(y, p['1.weight'], p['1.bias'])
    w = p['weight']
    if w.shape[-1] == 4:
        return F.conv2d(x, w, p['bias'], 2, 1)
    return F.conv2d(x, w, p['bias'], 1, 1)


def _upsample(p, x):
    """CFG:106-107 nearest x2 + 3x3 conv, or the last stage's plain 3x3 (CFG:394)."""
    if '1.weight' in p:
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        return F.conv2d(x, p['1.weight'], p['1.bias'], 1, 1)
    return F.conv2d(x, p['weight'], p['bias'], 1, 1)


def _num_stages(sd):
    return 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('downs.'))


# --------------------------------------------------------------------------- N10
def _trunk(sd, x, cond, groups, taps=None):
    """shared down/mid/up trunk, CFG:432-466 / DDP:413-447."""
    def tap(name, v):
        if taps is not None:
            taps[name] = v
    x = F.conv2d(x, sd['init_conv.weight'], sd['init_conv.bias'], 1, 3)
    tap('init_conv', x)
    r = x.clone()
    hs = []
    ns = _num_stages(sd)
    for i in range(ns):
        x = resnet_block(_sub(sd, f'downs.{i}.0'), x, groups, cond)
        tap(f'downs.{i}.0', x)
        hs.append(x)
        x = resnet_block(_sub(sd, f'downs.{i}.1'), x, groups, cond)
        tap(f'downs.{i}.1', x)
        x = _res_prenorm(_sub(sd, f'downs.{i}.2'), x, linear_attention)
        tap(f'downs.{i}.2', x)
        hs.append(x)
        x = _downsample(_sub(sd, f'downs.{i}.3'), x)
        tap(f'downs.{i}.3', x)
    x = resnet_block(_sub(sd, 'mid_block1'), x, groups, cond)
    tap('mid_block1', x)
    x = _res_prenorm(_sub(sd, 'mid_attn'), x, attention)
    tap('mid_attn', x)
    x = resnet_block(_sub(sd, 'mid_block2'), x, groups, cond)
    tap('mid_block2', x)
    for i in range(ns):
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(_sub(sd, f'ups.{i}.0'), x, groups, cond)
        tap(f'ups.{i}.0', x)
        x = torch.cat((x, hs.pop()), dim=1)
        x = resnet_block(_sub(sd, f'ups.{i}.1'), x, groups, cond)
        tap(f'ups.{i}.1', x)
        x = _res_prenorm(_sub(sd, f'ups.{i}.2'), x, linear_attention)
        tap(f'ups.{i}.2', x)
        x = _upsample(_sub(sd, f'ups.{i}.3'), x)
        tap(f'ups.{i}.3', x)
    x = torch.cat((x, r), dim=1)
    x = resnet_block(_sub(sd, 'final_res_block'), x, groups, cond)
    tap('final_res_block', x)
    return F.conv2d(x, sd['final_conv.weight'], sd['final_conv.bias'])


def cfg_unet_forward(sd, x, time, classes, rgb_flow, mask, keep_mask, groups=8, taps=None):
    """CFG Unet.forward, CFG:412-466.

    ``keep_mask``: (B,) bool — the class-dropout draw of CFG:422 made explicit
    (None == cond_drop_prob 0; all-False == the null pass of CFG:409)."""
    dim = sd['classes_emb.weight'].shape[1]
    c = class_mlp(sd, classes, keep_mask)
    x = torch.cat((x, rgb_flow * mask), dim=1)
    t = time_mlp(sd, time, dim)
    cond = torch.cat((t, c), dim=-1)
    return _trunk(sd, x, cond, groups, taps)


def cfg_unet_forward_with_cond_scale(sd, x, time, classes, rgb_flow, mask, keep_mask,
                                     cond_scale, groups=8):
    """Unet.forward_with_cond_scale, CFG:403-410."""
    logits = cfg_unet_forward(sd, x, time, classes, rgb_flow, mask, keep_mask, groups)
    if cond_scale == 1:
        return logits
    null = cfg_unet_forward(sd, x, time, classes, rgb_flow, mask,
                            torch.zeros(x.shape[0], dtype=torch.bool), groups)
    return null + (logits - null) * cond_scale


def ddp_unet_forward(sd, x, time, x_self_cond=None, self_condition=False, groups=8, taps=None):
    """DDP Unet.forward, DDP:408-447."""
    dim = sd['time_mlp.1.weight'].shape[1]
    if self_condition:
        if x_self_cond is None:
            x_self_cond = torch.zeros_like(x)
        x = torch.cat((x_self_cond, x), dim=1)
    t = time_mlp(sd, time, dim)
    return _trunk(sd, x, t, groups, taps)

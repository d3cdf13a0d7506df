"""Tensor-level wrappers of the C ABI (include/dmhomo_hip.h).

PyTorch is used here only to allocate device memory (``torch.empty``) and to carry
pointers; every value is produced by a kernel of libdmhomo_hip.so.  Activations
are NHWC fp32 tensors shaped (B, H, W, C).
"""
import ctypes as C

import os

import torch

from . import _lib
from ._lib import call, ptr, lib

F32 = torch.float32
# DMH_UP_SUBPIXEL=0: keep Upsample + conv3x3 as one 3x3 conv over the (virtually) upsampled input
SUBPIXEL_UP = __import__('os').environ.get('DMH_UP_SUBPIXEL', '1') != '0'


def _empty(shape, like, dtype=F32):
    return torch.empty(shape, device=like.device, dtype=dtype)


# ------------------------------------------------------------------ weight preparation
def ws_standardize(w, eps=1e-5):
    """N1: (w - mean_o) * rsqrt(var_o + eps) per output channel (CFG:123-126)."""
    w = w.detach().contiguous()
    out = torch.empty_like(w)
    cout = w.shape[0]
    call('dmh_ws_standardize', ptr(w), ptr(out), cout, w.numel() // cout, float(eps))
    return out


class PackBatch:
    """the weights a training step re-packs after every optimiser update, as ONE table for dmh_pack_conv_weights_multi
    (three kernel phases over all of them) + the few that need their own launches (``specials``: closures that pack into
    buffers allocated once).  Buffers and source pointers are fixed when the batch is built; ``run()`` is graph-capturable."""

    def __init__(self):
        self.jobs, self.pre, self.specials, self._keep, self._arr = [], [], [], [], None

    def add(self, src, ws, wpack, cout, c0, c1, kh, transposed):
        self.jobs.append(_lib.DmhPackJob(ptr(src), ptr(ws), ptr(wpack), cout, c0, c1, kh, int(transposed)))
        self._keep.append((src, ws, wpack))
        self._arr = None

    def run(self, eps=1e-5):
        for f in self.pre:                 # (weights that are first assembled from the parameters: padded, re-indexed)
            f()
        if self.jobs:
            if self._arr is None:
                self._arr = (_lib.DmhPackJob * len(self.jobs))(*self.jobs)
            call('dmh_pack_conv_weights_multi', C.cast(self._arr, C.c_void_p), len(self.jobs), float(eps))
        for f in self.specials:
            f()


def f16x3_default():
    """the fp16-piece kernels (DMH_CONV3_VARIANT unset or 9) serve the 1x1 / 3x3 convolutions"""
    return os.environ.get('DMH_CONV3_VARIANT', '9') == '9'


def _batchable(kh, stride, upsample2):
    # (dmh_pack_conv_weights_multi makes the fp16-piece images of the default kernels only)
    return kh in (1, 3) and stride == 1 and not upsample2 and os.environ.get('DMH_CONV3_VARIANT', '9') == '9'


class PackedConv:
    """a conv weight in dmh_conv2d's tile-major layout + its geometry.
    batch (PackBatch): the image is NOT made here but by ``batch.run()`` — from ``w_oihw`` (whose storage must stay where
    it is), or, with ``ws_from`` (the raw weight), from its standardised form, which the batch writes into ``w_oihw``."""
    __slots__ = ('wpack', 'bias', 'cout', 'c0', 'c1', 'k', 'stride', 'upsample2', '_w', '_up2')

    def __init__(self, w_oihw, bias, c0, c1=0, stride=1, upsample2=0, subpixel=True, batch=None, ws_from=None):
        w = w_oihw.detach()
        assert w.is_contiguous()
        cout, cin, kh, kw = w.shape
        assert cin == c0 + c1 and kh == kw, (w.shape, c0, c1)
        self._w = w
        n_up2 = lib().dmh_conv_up2_pack_floats(cout, c0) if (upsample2 and kh == 3 and c1 == 0 and subpixel and SUBPIXEL_UP) else -1
        self._up2 = n_up2 > 0
        if self._up2:                     # Upsample + conv3x3 as four 2x2 sub-pixel convs (upsample2 = 2)
            self.wpack = _empty((n_up2,), w)
            upsample2 = 2
        else:
            self.wpack = _empty((lib().dmh_conv_pack_floats(cout, c0, c1, kh, kw),), w)
        self.bias = None if bias is None else bias.detach().contiguous()
        self.cout, self.c0, self.c1, self.k, self.stride, self.upsample2 = cout, c0, c1, kh, stride, upsample2
        if batch is not None and _batchable(kh, stride, upsample2):
            if ws_from is not None:
                batch.add(ws_from.detach(), w, self.wpack, cout, c0, c1, kh, 0)
            else:
                batch.add(w, None, self.wpack, cout, c0, c1, kh, 0)
        else:
            assert ws_from is None
            if batch is not None:
                batch.specials.append(self.repack)
            self.repack()

    def repack(self):
        """(re)make the image from the weight tensor's current values"""
        if self._up2:
            call('dmh_pack_conv_weight_up2', ptr(self._w), ptr(self.wpack), self.cout, self.c0)
        else:
            call('dmh_pack_conv_weight', ptr(self._w), ptr(self.wpack), self.cout, self.c0, self.c1, self.k, self.k)


def conv_out_hw(pc, h, w):
    if pc.upsample2:
        return h * 2, w * 2
    if pc.stride == 2:
        return h // 2, w // 2
    return h, w


# launch timing of the conv kernels with HIP events on the launch stream (bench.py's roofline leg):
# when a list, every dmh_conv2d launch appends (start_event, end_event, k, stride, B, Hout, Wout, Cin, Cout, upsample2,
# has GroupNorm+SiLU prologue)
CONV_LOG = None


def rows_from_keep(keep, extra=0):
    """the active-row list of a classifier-free-guidance pass (include/dmhomo_hip.h, "Row subsets"): int32 (1 + B + extra,)
    = [n, the rows with keep != 0 ..., B .. B + extra - 1, (unused)]; every batched wrapper below takes it as ``rows=``."""
    B = keep.shape[0]
    rows = torch.empty((1 + B + extra,), device=keep.device, dtype=torch.int32)
    call('dmh_rows_from_keep', ptr(keep, torch.uint8), B, int(extra), ptr(rows, torch.int32))
    return rows


def _rows(rows):
    return ptr(rows, torch.int32)


def conv2d(pc, src0, src1=None, in_coef=None, res=None, res_coef=None, want_stats=False, in_bound=None, final=None,
           keep_out=True, pixel_stats=False, eps=1e-5, fin_out=None, rows=None):
    """K1/K2.  src0 (B,H,W,C0) [+ src1 (B,H,W,C1) = fused channel concat]. Returns out or (out, stats).
    in_bound (B, k), with in_coef: upper bounds of the prologue's |a*x+b| per sample (gn_finalize(want_bound=True)).
    final = (w (n, Cout), b (n,) or None), 1x1 convs with Cout <= 64 only: also apply that pointwise projection to every
    finished output pixel and return (out, y) with y (B, n, H, W) NCHW — ``final_conv_nchw(out, w, b)`` without the second
    pass over ``out`` (DmhConv.fin_*); keep_out=False: ``out`` itself is not stored (returned as None).
    fin_out: where y goes (a contiguous (B, n, H, W) tensor, e.g. a row slice of the caller's result) instead of a new tensor.
    pixel_stats (1x1 convs with Cout == 64): return (out, pstats) with pstats (B, H*W, 2) = the channel-LayerNorm (mean, rstd)
    of every output pixel, as ``dmh_pixel_stats(out)`` gives them, for ``linear_attention_fused(..., stats=)``."""
    B, H, W, c0 = src0.shape
    assert c0 == pc.c0 and (pc.c1 == 0) == (src1 is None), (src0.shape, pc.c0, pc.c1)
    if src1 is not None:
        assert src1.shape == (B, H, W, pc.c1), (src1.shape, pc.c1)
    ho, wo = conv_out_hw(pc, H, W)
    assert keep_out or final is not None
    out = _empty((B, ho, wo, pc.cout), src0) if keep_out else None
    fin_w = fin_b = None
    if final is not None:
        fin_w, fin_b = final
        assert fin_w.shape[1] == pc.cout and fin_w.is_contiguous() and not want_stats, (fin_w.shape, pc.cout)
        if fin_out is None:
            fin_out = _empty((B, fin_w.shape[0], ho, wo), src0)
        assert fin_out.shape == (B, fin_w.shape[0], ho, wo) and fin_out.is_contiguous() and fin_out.dtype == F32
    else:
        assert fin_out is None
    pst = None
    if pixel_stats:
        assert final is None and not want_stats and pc.cout == 64 and pc.k == 1, (pc.cout, pc.k)
        pst = _empty((B, ho * wo, 2), src0)
    stats = None
    if want_stats:
        tiles = lib().dmh_conv_tiles(ho, wo, pc.k, pc.stride)
        stats = _empty((B, tiles, pc.cout, 2), src0)
    if res is not None:
        assert res.shape == (B, ho, wo, pc.cout), (res.shape, (B, ho, wo, pc.cout))
    d = _lib.DmhConv(C.sizeof(_lib.DmhConv), ptr(src0), ptr(src1), ptr(pc.wpack), ptr(pc.bias), ptr(in_coef), ptr(res), ptr(res_coef),
                     ptr(out), ptr(stats), B, H, W, pc.c0, pc.c1, pc.cout, pc.k, pc.k, pc.stride, pc.upsample2,
                     ptr(in_bound), 0 if in_bound is None else in_bound.shape[1],
                     0 if fin_w is None else fin_w.shape[0], ptr(fin_w), ptr(fin_b), ptr(fin_out), ptr(pst), float(eps), _rows(rows))
    if CONV_LOG is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call('dmh_conv2d', C.byref(d))
        e1.record()
        CONV_LOG.append((e0, e1, pc.k, pc.stride, B, ho, wo, pc.c0 + pc.c1, pc.cout, pc.upsample2, in_coef is not None))
    else:
        call('dmh_conv2d', C.byref(d))
    if final is not None:
        return out, fin_out
    if pixel_stats:
        return out, pst
    return (out, stats) if want_stats else out


# ------------------------------------------------------------------ normalisation glue
# development knob (A/B runs): DMH_CONV_STATIC_BOUND=0 lets the fp16-piece convs search their staged tiles for the block
# maximum even where the producer's GroupNorm statistics bound it
STATIC_BOUND = os.environ.get('DMH_CONV_STATIC_BOUND', '1') != '0'


def gn_finalize(stats, gamma, beta, hw, groups, ss=None, eps=1e-5, want_bound=False, rows=None):
    """N2: stats (B,tiles,C,2) -> coef (B,2,C).  ss: (B, >=2C) view whose row b starts with (scale[C], shift[C]).
    want_bound: -> (coef, bound) with bound (B, groups) >= |a*x + b| over each (sample, group): ``conv2d(in_bound=)``."""
    B, tiles, Cc, _ = stats.shape
    coef = _empty((B, 2, Cc), stats)
    ss_ptr, ss_stride = None, 0
    if ss is not None:
        assert ss.stride(1) == 1 and ss.shape[0] == B and ss.shape[1] == 2 * Cc
        ss_ptr, ss_stride = C.c_void_p(ss.data_ptr()), ss.stride(0)
    if want_bound:
        bound = _empty((B, groups), stats)
        call('dmh_gn_finalize_bound', ptr(stats), tiles, ptr(gamma), ptr(beta), ss_ptr, ss_stride, ptr(coef), ptr(bound),
             B, Cc, groups, hw, float(eps), _rows(rows))
        return coef, bound
    call('dmh_gn_finalize', ptr(stats), tiles, ptr(gamma), ptr(beta), ss_ptr, ss_stride, ptr(coef), B, Cc, groups,
         hw, float(eps), _rows(rows))
    return coef


# widths whose ResnetBlock epilogue can hand the LayerNorm statistics to the LinearAttention behind it
# (development knob DMH_FUSED_PIXEL_STATS=0: the standalone dmh_pixel_stats pass everywhere, for A/B runs)
PIXEL_STATS_FUSABLE = () if os.environ.get('DMH_FUSED_PIXEL_STATS') == '0' else (64, 128, 256)


def gn_silu_residual(y, coef, res, pixel_stats=False, eps=1e-5, rows=None):
    """SiLU(a*y+b) + res.  pixel_stats: also return the (B, H*W, 2) per-pixel (mean, rstd) of the channel LayerNorm of the
    result — what ``dmh_pixel_stats`` would compute from it — for ``linear_attention_fused(..., stats=)``."""
    B, H, W, Cc = y.shape
    out = torch.empty_like(y)
    if pixel_stats:
        stats = _empty((B, H * W, 2), y)
        call('dmh_gn_silu_residual_stats', ptr(y), ptr(coef), ptr(res), ptr(out), ptr(stats), B, H * W, Cc, float(eps),
             _rows(rows))
        return out, stats
    call('dmh_gn_silu_residual', ptr(y), ptr(coef), ptr(res), ptr(out), B, H * W, Cc, _rows(rows))
    return out


def chan_layernorm(x, g, res=None, eps=1e-5, rows=None):
    """N4 (+ optional residual add)."""
    Cc = x.shape[-1]
    out = torch.empty_like(x)
    npix = x.numel() // Cc
    call('dmh_chan_layernorm', ptr(x), ptr(g), ptr(res), ptr(out), npix, Cc, float(eps), _rows(rows), npix // x.shape[0])
    return out


# ------------------------------------------------------------------ attention cores
def linear_attention_core(qkv, scale, rows=None):
    """K3.  qkv (B,H,W,384) -> (B,H,W,128)."""
    B, H, W, c = qkv.shape
    assert c == 384
    n = H * W
    partial = _empty((lib().dmh_linattn_partial_floats(B, n),), qkv)
    ctx = _empty((B, 4, 32, 32), qkv)
    out = _empty((B, H, W, 128), qkv)
    call('dmh_linattn_context', ptr(qkv), ptr(partial), B, n, _rows(rows))
    call('dmh_linattn_merge', ptr(partial), ptr(ctx), B, n, _rows(rows))
    call('dmh_linattn_apply', ptr(qkv), ptr(ctx), ptr(out), B, n, float(scale), _rows(rows))
    return out


class PackedLinAttn:
    """to_qkv weight of a LinearAttention (384, C, 1, 1) packed for the fused kernels (once per weight version)."""

    def __init__(self, w_qkv):
        c = w_qkv.shape[1]
        assert w_qkv.shape[0] == 384 and c % 32 == 0, w_qkv.shape
        w = w_qkv.detach().reshape(384, c).contiguous().float()
        self.c = c
        self.wpack = _empty((lib().dmh_linattn_fused_pack_floats(c),), w)
        call('dmh_linattn_fused_pack', ptr(w), ptr(self.wpack), c)


class PackedLinAttnOut:
    """to_out of a LinearAttention with dim 64: conv weight (64, 128, 1, 1) + bias + the gain of its LayerNorm, packed for
    the fully fused second pass."""

    def __init__(self, w_out, bias, ln_g):
        assert tuple(w_out.shape[:2]) == (64, 128), w_out.shape
        w = w_out.detach().reshape(64, 128).contiguous().float()
        self.wpack = _empty((lib().dmh_linattn_out_pack_floats(),), w)
        call('dmh_linattn_out_pack', ptr(w), ptr(self.wpack))
        self.bias = bias.detach().contiguous().float()
        self.ln_g = ln_g.detach().reshape(-1).contiguous().float()


def linear_attention_fused(x, ln_g, pla, scale, eps=1e-5, out=None, stats=None, rows=None):
    """K3f.  x (B,H,W,C) -> attention core output (B,H,W,128) of LinearAttention(PreNorm-LayerNorm(x)): LayerNorm,
    to_qkv and both attention passes in two kernels, q/k/v never stored.
    With ``out`` (PackedLinAttnOut, C == 64) the second pass also applies to_out, its LayerNorm and the residual:
    returns x + LN(to_out(core)) of shape (B,H,W,64) — the whole Residual(PreNorm(LinearAttention)) block."""
    B, H, W, c = x.shape
    assert c == pla.c
    n = H * W
    if stats is None:     # (the producer of x may have written them already: gn_silu_residual(pixel_stats=True))
        stats = _empty((B, n, 2), x)
        call('dmh_pixel_stats', ptr(x), ptr(stats), B * n, c, float(eps), _rows(rows), n)
    assert stats.shape == (B, n, 2)
    ns = lib().dmh_linattn_fused_splits(B, n)
    partial = _empty((B, ns, 4, 1088), x)
    ctx = _empty((B, 4, 32, 32), x)
    call('dmh_linattn_fused_context', ptr(x), ptr(stats), ptr(ln_g), ptr(pla.wpack), ptr(partial), B, n, c, _rows(rows))
    call('dmh_linattn_merge_n', ptr(partial), ptr(ctx), B, n, ns, _rows(rows))
    if out is not None:
        y = _empty((B, H, W, 64), x)
        call('dmh_linattn_fused_apply_out', ptr(x), ptr(stats), ptr(ln_g), ptr(pla.wpack), ptr(ctx), ptr(out.wpack),
             ptr(out.bias), ptr(out.ln_g), ptr(y), B, n, c, float(scale), float(eps), _rows(rows))
        return y
    o = _empty((B, H, W, 128), x)
    call('dmh_linattn_fused_apply', ptr(x), ptr(stats), ptr(ln_g), ptr(pla.wpack), ptr(ctx), ptr(o), B, n, c,
         float(scale), _rows(rows))
    return o


def attention_core(qkv, scale, rows=None):
    """K4.  qkv (B,H,W,384) -> (B,H,W,128)."""
    B, H, W, c = qkv.shape
    assert c == 384
    out = _empty((B, H, W, 128), qkv)
    call('dmh_attention', ptr(qkv), ptr(out), B, H * W, float(scale), _rows(rows))
    return out


# ------------------------------------------------------------------ embeddings
ACT = {None: 0, 'silu': 1, 'gelu': 2}


def sinusoidal_embed(t, freq):
    R, dim = t.shape[0], freq.shape[0] * 2
    out = _empty((R, dim), freq)
    call('dmh_sinusoidal_embed', ptr(t, torch.int64), ptr(freq), ptr(out), R, dim)
    return out


def fourier_embed(t, weights):
    """RandomOrLearnedSinusoidalPosEmb CFG:185-190: (R,) int64, (half,) -> (R, 2*half + 1)"""
    R, half = t.shape[0], weights.shape[0]
    out = _empty((R, 2 * half + 1), weights)
    call('dmh_fourier_embed', ptr(t, torch.int64), ptr(weights), ptr(out), R, half)
    return out


def class_embed(classes, keep, table, null_emb):
    R, dim = classes.shape[0], table.shape[1]
    out = _empty((R, dim), table)
    call('dmh_class_embed', ptr(classes, torch.int64), ptr(keep, torch.uint8), ptr(table), ptr(null_emb), ptr(out), R,
         dim, table.shape[0])
    return out


def linear(x, wt, bias, act_in=None, act_out=None, out=None):
    """y = act_out(act_in(x) @ wt + bias); wt is W^T (in, out). x / out may be column slices (unit inner stride)."""
    R, in_dim = x.shape
    out_dim = wt.shape[1]
    assert wt.shape[0] == in_dim and x.stride(1) == 1
    if out is None:
        out = _empty((R, out_dim), wt)
    assert out.shape == (R, out_dim) and out.stride(1) == 1
    call('dmh_linear', C.c_void_p(x.data_ptr()), x.stride(0), ptr(wt), ptr(bias), C.c_void_p(out.data_ptr()),
         out.stride(0), R, in_dim, out_dim, ACT[act_in], ACT[act_out])
    return out


def ss_gather(T, Ct, bias, cursor, classes, keep, out=None):
    """(scale, shift) rows of a replayed denoise step from the tables of ``UnetEngine.ss_tables``:
    out[b] = (T[cursor] + Ct[keep[b] ? classes[b] : last row]) + bias — bitwise ``linear`` over the full embedding."""
    B, N = classes.shape[0], T.shape[1]
    if out is None:
        out = _empty((B, N), T)
    assert out.shape == (B, N) and out.is_contiguous() and Ct.shape[1] == N and bias.shape == (N,)
    call('dmh_ss_gather', ptr(T), ptr(Ct), ptr(bias), ptr(cursor, torch.int32), ptr(classes, torch.int64),
         ptr(keep, torch.uint8), Ct.shape[0] - 1, ptr(out), B, N, T.shape[0])
    return out


# ------------------------------------------------------------------ sampler glue
def assemble_input(a, b=None, m=None, reps=1, cpad=None):
    """(B,Ca,H,W) [+ (B,Cb,H,W) * (B,1,H,W)] NCHW -> NHWC (reps*B, H, W, cpad), zero padded."""
    B, ca, H, W = a.shape
    cb = 0 if b is None else b.shape[1]
    if cpad is None:
        cpad = (ca + cb + 3) // 4 * 4
    out = _empty((reps * B, H, W, cpad), a)
    call('dmh_assemble_input', ptr(a), ca, ptr(b), cb, ptr(m), ptr(out), B, reps, H * W, cpad)
    return out


def final_conv_nchw(x, w, bias):
    R, H, W, Cc = x.shape
    cout = w.shape[0]
    out = _empty((R, cout, H, W), x)
    call('dmh_final_conv_nchw', ptr(x), ptr(w), ptr(bias), ptr(out), R, H * W, Cc, cout)
    return out


OBJECTIVE = {'pred_noise': 0, 'pred_x0': 1, 'pred_v': 2}
MODE_DDIM, MODE_LAST, MODE_DDPM = 0, 1, 2


def sampler_step(step, model_cond, model_null, x, noise, want_x_start=True, want_pred_noise=False, keep=None):
    """keep (B,) uint8, with model_null: rows with keep == 0 take model_null as their conditional logits (rows the conditional
    pass did not compute: Unet.dedup_dropped_rows)"""
    img = torch.empty_like(x)
    xs = torch.empty_like(x) if want_x_start else None
    pn = torch.empty_like(x) if want_pred_noise else None
    call('dmh_sampler_step', C.byref(step), ptr(model_cond), ptr(model_null), ptr(x), ptr(noise), ptr(img), ptr(xs),
         ptr(pn), x.numel(), ptr(keep, torch.uint8), x.numel() // x.shape[0])
    return img, xs, pn


def step_table(steps, times, device):
    """device tables of a replayed sampling loop: steps (list of DmhStep, host-computed) -> (uint8 tensor holding the
    packed structs, int64 times, int32 cursor, one-struct uint8 'current' buffer)."""
    n = len(steps)
    # the kernels read these entries from device memory and cannot validate them at launch: do it here
    if n < 1 or len(times) != n:
        raise ValueError(f'step_table: {n} steps, {len(times)} times')
    for i, st in enumerate(steps):
        if st.objective not in (0, 1, 2) or st.mode not in (MODE_DDIM, MODE_LAST, MODE_DDPM):
            raise ValueError(f'step_table: entry {i} has objective {st.objective} / mode {st.mode}')
        if (st.mode == MODE_LAST) != (i == n - 1) and st.mode != MODE_DDPM:
            raise ValueError(f'step_table: entry {i} of {n} has mode {st.mode}: MODE_LAST belongs to the last entry only')
    sz = C.sizeof(_lib.DmhStep)
    arr = (_lib.DmhStep * n)(*steps)
    raw = torch.frombuffer(bytearray(C.string_at(C.addressof(arr), n * sz)), dtype=torch.uint8).clone()
    table = raw.to(device)
    tt = torch.tensor(list(times), dtype=torch.int64).to(device)
    cursor = torch.zeros((1,), dtype=torch.int32, device=device)
    cur = torch.zeros((sz,), dtype=torch.uint8, device=device)
    return table, tt, cursor, cur


def sampler_seek(cursor, k, table, times, cur, tcond):
    """k >= 0: cursor = k; k < 0: cursor += 1 (clamped); cur = table[cursor]; tcond[:] = times[cursor]."""
    call('dmh_sampler_seek', ptr(cursor, torch.int32), int(k), ptr(table, torch.uint8), ptr(times, torch.int64),
         times.shape[0], ptr(cur, torch.uint8), ptr(tcond, torch.int64), tcond.shape[0])


def sampler_step_dev(cur, model_cond, model_null, x, noise, out=None, keep=None):
    """sampler_step with its DmhStep in device memory (``cur`` of step_table); out may be x (in place)."""
    img = torch.empty_like(x) if out is None else out
    call('dmh_sampler_step_dev', ptr(cur, torch.uint8), ptr(model_cond), ptr(model_null), ptr(x), ptr(noise), ptr(img),
         None, None, x.numel(), ptr(keep, torch.uint8), x.numel() // x.shape[0])
    return img


def rng_indexed(shape, sample_ids, state, kind=0):
    """one draw of the sample-indexed generator (dmh_rng_indexed): (B, *shape[1:]) fp32 whose row b is a pure function of
    (state[0] = seed, sample_ids[b], state[1] = draw index, element); the launch advances the draw index.
    kind 0: N(0,1) (CFG:679,705), 1: uniform [0,1) (CFG:90), 2: the raw Philox words (bit patterns; tests)."""
    B = int(shape[0])
    assert sample_ids.shape == (B,) and state.shape == (4,) and state.dtype == torch.int64
    out = torch.empty(tuple(shape), device=sample_ids.device, dtype=F32)
    call('dmh_rng_indexed', ptr(out), B, out.numel() // B, ptr(sample_ids, torch.int64), ptr(state, torch.int64), int(kind))
    return out


def rng_keep_mask(sample_ids, state, prob):
    """(B,) uint8: uniform draw of the indexed generator < prob (the class-dropout mask of CFG:84-90), one launch."""
    B = sample_ids.shape[0]
    out = torch.empty((B,), device=sample_ids.device, dtype=torch.uint8)
    call('dmh_rng_keep_mask', ptr(out, torch.uint8), B, ptr(sample_ids, torch.int64), ptr(state, torch.int64), float(prob))
    return out


def affine(x, scale, shift, out=None):
    x = x.contiguous()
    y = torch.empty_like(x) if out is None else out
    call('dmh_affine', ptr(x), ptr(y), float(scale), float(shift), x.numel())
    return y


def lerp(a, b, lam):
    """(1 - lam) * a + lam * b  (DDP:746) on the q_sample kernel."""
    B = a.shape[0]
    ca = torch.full((B,), 1.0 - lam, device=a.device, dtype=torch.float32)
    cb = torch.full((B,), lam, device=a.device, dtype=torch.float32)
    return q_sample(a.contiguous(), b.contiguous(), ca, cb)


def affine_tail_(x, c0, scale, shift):
    B, Cc, H, W = x.shape
    call('dmh_affine_tail', ptr(x), B, Cc, H * W, c0, float(scale), float(shift))
    return x


def q_sample(x_start, noise, ca, cb):
    out = torch.empty_like(x_start)
    B = x_start.shape[0]
    call('dmh_q_sample', ptr(x_start), ptr(noise), ptr(ca), ptr(cb), ptr(out), B, x_start.numel() // B)
    return out


def rows_lincomb(x, ca, y=None, cb=None, div=None, clamp=False):
    """out = ca[b] * x (+ cb[b] * y) (/ div[b]) (clamped to [-1, 1]); ca / cb / div: (B,) fp32 per-sample coefficients
    (``extract`` of a schedule buffer at a per-row timestep): D4 / D7 of SURVEY 8a in the reference's op order."""
    x = x.contiguous()
    B = x.shape[0]
    out = torch.empty_like(x)
    call('dmh_rows_lincomb', ptr(x), ptr(None if y is None else y.contiguous()), ptr(ca.contiguous()),
         ptr(None if cb is None else cb.contiguous()), ptr(None if div is None else div.contiguous()), ptr(out), B,
         x.numel() // B, int(bool(clamp)))
    return out


def diff_mean(a, b, mask=None, squared=False):
    """per-sample mean over (C,H,W) of mask * |a-b| (or (a-b)^2): the loss terms of p_losses (CFG:796-802)."""
    B, Cc, H, W = a.shape
    ws = torch.empty((B, 64), device=a.device, dtype=torch.float64)
    out = torch.empty((B,), device=a.device, dtype=F32)
    call('dmh_diff_mean', ptr(a), ptr(b), ptr(mask), int(bool(squared)), ptr(ws, torch.float64), ptr(out), B, Cc,
         H * W)
    return out


def loss_combine(l, photo, w):
    out = torch.empty((1,), device=l.device, dtype=F32)
    call('dmh_loss_combine', ptr(l), ptr(photo), ptr(w), ptr(out), l.shape[0])
    return out[0]


def to_uint8(img):
    out = torch.empty(img.shape, device=img.device, dtype=torch.uint8)
    call('dmh_to_uint8', ptr(img), ptr(out, torch.uint8), img.numel())
    return out


# ------------------------------------------------------------------ geometry
def homography_flow(Hm, H, W, max_flow=256., want_flow=True, want_rgb=True):
    """G2+G3.  Hm (B,3,3) f64 device -> flow (B,2,H,W), rgb (B,3,H,W)."""
    B = Hm.shape[0]
    flow = torch.empty((B, 2, H, W), device=Hm.device, dtype=F32) if want_flow else None
    rgb = torch.empty((B, 3, H, W), device=Hm.device, dtype=F32) if want_rgb else None
    call('dmh_homography_flow', ptr(Hm, torch.float64), ptr(flow), ptr(rgb), B, H, W, float(max_flow))
    return flow, rgb


def flow_to_image(flow, max_flow=256.):
    """G3: flow (B,2,H,W) -> rgb (B,3,H,W)."""
    B, _, H, W = flow.shape
    rgb = torch.empty((B, 3, H, W), device=flow.device, dtype=F32)
    call('dmh_flow_to_image', ptr(flow), ptr(rgb), B, H * W, float(max_flow))
    return rgb


FLOW_WARP_PAD = {'border': 0, 'zeros': 1, 'reflection': 2}
FLOW_WARP_MODE = {'bilinear': 0, 'nearest': 1, 'bicubic': 2}


def flow_warp(x, flow, want_indices=False, pad='border', mode='bilinear'):
    B, Cc, H, W = x.shape
    assert flow.shape == (B, 2, H, W)
    out = torch.empty_like(x)
    x0 = y0 = None
    if want_indices:
        x0 = torch.empty((B, H, W), device=x.device, dtype=torch.int32)
        y0 = torch.empty((B, H, W), device=x.device, dtype=torch.int32)
    call('dmh_flow_warp', ptr(x), ptr(flow), ptr(out), ptr(x0, torch.int32), ptr(y0, torch.int32), B, Cc, H, W,
         FLOW_WARP_PAD[pad], FLOW_WARP_MODE[mode])
    return (out, x0, y0) if want_indices else out


def pixel_grid(B, H, W, start, device):
    """get_grid DDP:1558-1574: (B,2,H,W) fp32 pixel coordinates (+ start)."""
    out = torch.empty((B, 2, H, W), device=device, dtype=F32)
    call('dmh_pixel_grid', ptr(out), B, H, W, float(start))
    return out


def norm_grid(v):
    """norm_grid DDP:1292-1299: (B,2,H,W) pixel coordinates -> (B,H,W,2) in [-1,1]."""
    B, two, H, W = v.shape
    assert two == 2
    out = torch.empty((B, H, W, 2), device=v.device, dtype=F32)
    call('dmh_norm_grid', ptr(v.contiguous()), ptr(out), B, H, W)
    return out


def homography_flow_points(Hm, idx):
    """get_flow_np DDP:927-969: Hm (B,divide,3,3) f64, idx (Bi,3,H,W) f64 -> flow (B,2,H,W) f64."""
    B, divide = Hm.shape[:2]
    Bi, three, H, W = idx.shape
    assert three == 3 and Hm.shape[2:] == (3, 3)
    out = torch.empty((B, 2, H, W), device=Hm.device, dtype=torch.float64)
    call('dmh_homography_flow_points', ptr(Hm.contiguous(), torch.float64), ptr(idx.contiguous(), torch.float64),
         ptr(out, torch.float64), B, divide, Bi, H, W)
    return out


def dlt_points(src, off):
    """DLT_solve DDP:1612-1643: src, off (N,P,2) f64 -> (N,3,3) f64 least-squares homographies."""
    N, P, two = src.shape
    assert two == 2 and off.shape == src.shape
    ws = torch.empty((N, _lib.DLT_BLOCKS, 44), device=src.device, dtype=torch.float64)
    out = torch.empty((N, 3, 3), device=src.device, dtype=torch.float64)
    call('dmh_dlt_points', ptr(src.contiguous(), torch.float64), ptr(off.contiguous(), torch.float64),
         ptr(ws, torch.float64), ptr(out, torch.float64), N, P)
    return out


def dlt_homography(flow):
    """G5: flow (B,2,H,W) fp32 -> (B,3,3) f64."""
    B, _, H, W = flow.shape
    ws = torch.empty((B, _lib.DLT_BLOCKS, 44), device=flow.device, dtype=torch.float64)
    out = torch.empty((B, 3, 3), device=flow.device, dtype=torch.float64)
    call('dmh_dlt_homography', ptr(flow), ptr(ws, torch.float64), ptr(out, torch.float64), B, H, W)
    return out


# ------------------------------------------------------------------ training (first pieces)
def conv_wgrad(dy, src0, src1=None, k=3, in_coef=None, want_bias=True, ups=0):
    """weight (and bias) gradient of the stride-1 kxk conv whose input was cat(src0, src1) (after the optional
    SiLU(a*src0+b) prologue; ups=1: after a nearest x2 upsampling; k=2: 'valid' conv over the space-to-depth view) and
    whose output gradient is dy.  NHWC in, OIHW (Cout, C0+C1, k, k) out."""
    B, H, W, cout = dy.shape
    c0 = src0.shape[3]
    c1 = 0 if src1 is None else src1.shape[3]
    dw = _empty((cout, c0 + c1, k, k), dy)
    db = _empty((cout,), dy) if want_bias else None
    work = _empty((lib().dmh_conv_wgrad_workspace_floats(B, H, W, c0, c1, cout, k),), dy)
    call('dmh_conv_wgrad', ptr(dy.contiguous()), ptr(src0), ptr(src1), ptr(in_coef), ptr(dw), ptr(db), ptr(work), B, H, W,
         c0, c1, cout, k, int(ups))
    return (dw, db) if want_bias else dw


class _DgradPacked(PackedConv):
    """the data-gradient conv of a stride-1 conv with weight w, its image made by a PackBatch from w itself (taps flipped
    and (Cout, Cin) exchanged by index arithmetic in the pack kernel: no flipped / transposed copy of the weight)"""
    __slots__ = ()

    def __init__(self, w, c_in_total, batch):
        w = w.detach()
        cout_src, cin_src, kh, kw = w.shape
        assert cin_src == c_in_total and kh == kw and w.is_contiguous()
        self._w, self._up2 = w, False
        self.wpack = _empty((lib().dmh_conv_pack_floats(cin_src, cout_src, 0, kh, kw),), w)
        self.bias = None
        self.cout, self.c0, self.c1, self.k, self.stride, self.upsample2 = cin_src, cout_src, 0, kh, 1, 0
        batch.add(w, None, self.wpack, cin_src, cout_src, 0, kh, 1)


def conv_dgrad_pack(w, c_in_total, batch=None):
    """PackedConv that computes the DATA gradient of a stride-1 conv with weight w (Cout, Cin, k, k): the same conv
    kernel on dy with the taps flipped and (Cout, Cin) transposed (no bias)."""
    if batch is not None and _batchable(w.shape[-1], 1, 0):
        return _DgradPacked(w, c_in_total, batch)
    wt = w.detach().flip(2, 3).transpose(0, 1).contiguous()          # (Cin, Cout, k, k)
    assert wt.shape[0] == c_in_total
    if batch is not None:            # (exact-fp32 variants: no table-driven pack) re-made from w on every batch.run()
        w_src = w.detach()
        batch.pre.append(lambda: wt.copy_(w_src.flip(2, 3).transpose(0, 1)))
    return PackedConv(wt, None, w.shape[0], batch=batch)


def gn_finalize_train(stats, gamma, beta, hw, groups, ss=None, eps=1e-5):
    """gn_finalize that also returns mr (B, groups, 2) = (mean, rstd), saved for gn_silu_backward."""
    B, tiles, Cc, _ = stats.shape
    coef = _empty((B, 2, Cc), stats)
    mr = _empty((B, groups, 2), stats)
    ss_ptr, ss_stride = None, 0
    if ss is not None:
        assert ss.stride(1) == 1 and ss.shape[0] == B and ss.shape[1] == 2 * Cc
        ss_ptr, ss_stride = C.c_void_p(ss.data_ptr()), ss.stride(0)
    call('dmh_gn_finalize_train', ptr(stats), tiles, ptr(gamma), ptr(beta), ss_ptr, ss_stride, ptr(coef), ptr(mr), B, Cc,
         groups, hw, float(eps))
    return coef, mr


def gn_silu_backward(dout, y, coef, mr, gamma, beta, groups, ss=None):
    """backward of GroupNorm -> (scale+1, shift) -> SiLU: -> dy (B,H,W,C), dgamma (C,), dbeta (C,), dss (B, 2C) = the
    gradient wrt the (scale, shift) row of the ResnetBlock mlp (zeros are NOT returned when ss is None: dss = None)."""
    B, H, W, Cc = y.shape
    hw = H * W
    dy = _empty(y.shape, y)
    pg = _empty((B, 4, Cc), y)
    part = _empty((B, lib().dmh_gn_bwd_chunks(hw), Cc, 2), y)
    bcoef = _empty((B, 3, Cc), y)
    ss_ptr, ss_stride = None, 0
    if ss is not None:
        assert ss.stride(1) == 1 and ss.shape[0] == B and ss.shape[1] == 2 * Cc
        ss_ptr, ss_stride = C.c_void_p(ss.data_ptr()), ss.stride(0)
    call('dmh_gn_silu_backward', ptr(dout), ptr(y), ptr(coef), ptr(mr), ptr(gamma), ptr(beta), ss_ptr, ss_stride, ptr(dy),
         ptr(pg), ptr(part), ptr(bcoef), B, hw, Cc, groups)
    # pg is (B, 4, C): per sample (dgamma, dbeta, dscale, dshift) parts.  All four are summed over the batch in place (the
    # last two sums are not used: cheaper than a strided copy of the first two in front of the kernel), and the (scale, shift)
    # gradient goes back as a VIEW of pg — its one consumer copies it into its slice of the mlp output's gradient
    gb = _empty((4, Cc), y)
    call('dmh_sum_over_batch', ptr(pg), ptr(gb), B, 4 * Cc)
    dss = pg.reshape(B, 4 * Cc)[:, 2 * Cc:] if ss is not None else None
    return dy, gb[0], gb[1], dss


def ws_backward(w, dwh, eps=1e-5):
    """gradient wrt the raw weight from the gradient wrt the standardised one (CFG:120-126)."""
    cout = w.shape[0]
    w, dwh = w.contiguous(), dwh.contiguous()
    dw = torch.empty_like(w)
    call('dmh_ws_backward', ptr(w), ptr(dwh), ptr(dw), cout, w.numel() // cout, float(eps))
    return dw


def chan_layernorm_backward(x, g, dout, eps=1e-5):
    """backward of chan_layernorm (without its residual input, whose gradient is dout itself) -> dx, dg."""
    Cc = x.shape[-1]
    x, dout = x.contiguous(), dout.contiguous()
    dx = torch.empty_like(x)
    nb = lib().dmh_lnb_blocks()
    part = _empty((nb, Cc), x)
    call('dmh_chan_layernorm_backward', ptr(x), ptr(g), ptr(dout), ptr(dx), ptr(part), x.numel() // Cc, Cc, float(eps))
    dg = _empty((Cc,), x)
    call('dmh_sum_over_batch', ptr(part), ptr(dg), nb, Cc)
    return dx, dg


def linear_attention_core_train(qkv, scale):
    """linear_attention_core that also returns what its backward needs: (out, saved)."""
    B, H, W, c = qkv.shape
    assert c == 384
    n = H * W
    partial = _empty((lib().dmh_linattn_partial_floats(B, n),), qkv)
    ctx = _empty((B, 4, 32, 32), qkv)
    ms = _empty((B, 4, 32, 2), qkv)
    out = _empty((B, H, W, 128), qkv)
    call('dmh_linattn_context', ptr(qkv), ptr(partial), B, n, None)
    call('dmh_linattn_merge_ms', ptr(partial), ptr(ctx), ptr(ms), B, n)
    call('dmh_linattn_apply', ptr(qkv), ptr(ctx), ptr(out), B, n, float(scale), None)
    return out, dict(qkv=qkv, ctx=ctx, ms=ms, scale=float(scale))


def linear_attention_core_backward(sv, dout):
    """gradient wrt qkv (B,H,W,384) of the LinearAttention core from the gradient wrt its output (B,H,W,128)."""
    qkv = sv['qkv']
    B, H, W, _ = qkv.shape
    n = H * W
    dqkv = torch.empty_like(qkv)
    work = _empty((lib().dmh_linattn_bwd_workspace_floats(B, n),), qkv)
    call('dmh_linattn_backward', ptr(qkv), ptr(sv['ctx']), ptr(sv['ms']), ptr(dout.contiguous()), ptr(dqkv), ptr(work), B, n,
         sv['scale'])
    return dqkv


def _strides4(*v):
    return (C.c_int64 * 4)(*v)


def bgemm(a, sa, b, sb, c, sc, M, N, K, nbo, nbi, alpha=1.0):
    """c[bo][bi] = alpha * a[bo][bi] (MxK) @ b[bo][bi] (KxN); sa/sb/sc = (outer batch, inner batch, row, column) strides in
    floats; a, b, c: tensors or (tensor, float offset) pairs."""
    def base(t):
        if isinstance(t, tuple):
            return C.c_void_p(t[0].data_ptr() + 4 * t[1])
        return ptr(t)
    call('dmh_bgemm', base(a), _strides4(*sa), base(b), _strides4(*sb), base(c), _strides4(*sc), M, N, K, nbo, nbi,
         float(alpha))


def attention_core_train(qkv, scale):
    """bottleneck Attention core (CFG:287-295) on the small-GEMM kernels, keeping P = softmax(sim): (out (B,H,W,128), saved)."""
    B, H, W, c = qkv.shape
    assert c == 384
    n = H * W
    S = _empty((B, 4, n, n), qkv)
    # sim[i][j] = scale * sum_d q[i][d] k[j][d]
    bgemm(qkv, (n * 384, 32, 384, 1), (qkv, 128), (n * 384, 32, 1, 384), S, (4 * n * n, n * n, n, 1), n, n, 32, B, 4, scale)
    P = torch.empty_like(S)
    call('dmh_softmax_rows', ptr(S), ptr(P), B * 4 * n, n)
    out = _empty((B, H, W, 128), qkv)
    # out[i][h*32+d] = sum_j P[i][j] v[j][d]
    bgemm(P, (4 * n * n, n * n, n, 1), (qkv, 256), (n * 384, 32, 384, 1), out, (n * 128, 32, 128, 1), n, 32, n, B, 4)
    return out, dict(qkv=qkv, P=P, scale=float(scale))


def attention_core_backward(sv, dout):
    """gradient wrt qkv (B,H,W,384) of the Attention core from the gradient wrt its output (B,H,W,128)."""
    qkv, P, scale = sv['qkv'], sv['P'], sv['scale']
    B, H, W, _ = qkv.shape
    n = H * W
    dout = dout.contiguous()
    dqkv = torch.empty_like(qkv)
    sP, sQ, sO = (4 * n * n, n * n, n, 1), (n * 384, 32, 384, 1), (n * 128, 32, 128, 1)
    # dv[j][d] = sum_i P[i][j] dout[i][d]
    bgemm(P, (4 * n * n, n * n, 1, n), dout, sO, (dqkv, 256), sQ, n, 32, n, B, 4)
    # dP[i][j] = sum_d dout[i][d] v[j][d]
    dP = torch.empty_like(P)
    bgemm(dout, sO, (qkv, 256), (n * 384, 32, 1, 384), dP, sP, n, n, 32, B, 4)
    call('dmh_softmax_rows_backward', ptr(P), ptr(dP), B * 4 * n, n)          # dP <- dsim
    # dq[i][d] = scale * sum_j dsim[i][j] k[j][d] ;  dk[j][d] = scale * sum_i dsim[i][j] q[i][d]
    bgemm(dP, sP, (qkv, 128), sQ, dqkv, sQ, n, 32, n, B, 4, scale)
    bgemm(dP, (4 * n * n, n * n, 1, n), qkv, sQ, (dqkv, 128), sQ, n, 32, n, B, 4, scale)
    return dqkv


def conv_up_backward(dy, x, w, dpack=None):
    """backward of Upsample = nearest x2 -> conv3x3 (CFG:106-107): dy (B,2H,2W,Cout), x (B,H,W,C) -> dx, dw, db."""
    B, H, W, c = x.shape
    dpack = dpack or conv_dgrad_pack(w, c)
    gup = conv2d(dpack, dy)                                             # gradient wrt the upsampled tensor
    dx = torch.empty_like(x)
    call('dmh_sumpool2', ptr(gup), ptr(dx), B, H, W, c)
    dw, db = conv_wgrad(dy, x, k=3, ups=1)
    return dx, dw, db


# Downsample 4x4 / stride 2 / pad 1: for input row parity r the taps (ky, row offset a) that reach it
_DOWN_K = {0: {-1: 3, 0: 1}, 1: {0: 2, 1: 0}}


def conv_down_dgrad_pack(w, batch=None):
    """PackedConv of the 3x3 conv over dy that yields the data gradient of the 4x4/stride-2 conv in pixel-shuffle
    layout: out[m][l][(ry*2+rx)*C + c] = dx[2m+ry][2l+rx][c]."""
    cout, c = w.shape[0], w.shape[1]
    w3 = torch.zeros((4 * c, cout, 3, 3), device=w.device, dtype=torch.float32)

    def fill():                            # (the taps that no parity reaches stay zero)
        for ry in (0, 1):
            for rx in (0, 1):
                par = ry * 2 + rx
                for a, ky in _DOWN_K[ry].items():
                    for b, kx in _DOWN_K[rx].items():
                        w3[par * c:(par + 1) * c, :, a + 1, b + 1] = w[:, :, ky, kx].t()
    fill()
    if batch is not None:
        batch.pre.append(fill)
    return PackedConv(w3, None, cout, batch=batch)


def conv_down_backward(dy, x, w, dpack=None):
    """backward of Downsample = conv 4x4 / stride 2 / pad 1 (CFG:110-111): dy (B,H/2,W/2,Cout), x (B,H,W,C) -> dx, dw, db"""
    B, H, W, c = x.shape
    cout = dy.shape[3]
    dpack = dpack or conv_down_dgrad_pack(w)
    g4 = conv2d(dpack, dy)                                              # (B, H/2, W/2, 4C)
    dx = torch.empty_like(x)
    call('dmh_d2s', ptr(g4), ptr(dx), B, H // 2, W // 2, c)
    X = _empty((B, H // 2 + 1, W // 2 + 1, 4 * c), x)
    call('dmh_s2d_shift', ptr(x.contiguous()), ptr(X), B, H, W, c)
    dw2, db = conv_wgrad(dy, X, k=2)                                    # (Cout, 4C, 2, 2): [(py,px,c)][dy][dx]
    dw = dw2.view(cout, 2, 2, c, 2, 2).permute(0, 3, 4, 1, 5, 2).reshape(cout, c, 4, 4).contiguous()
    return dx, dw, db


def act(x, mode, dy=None):
    """mode 'silu' | 'gelu': f(x), or dy * f'(x) when dy is given."""
    x = x.contiguous()
    out = torch.empty_like(x)
    call('dmh_act', ptr(x), ptr(None if dy is None else dy.contiguous()), ptr(out), x.numel(), {'silu': 1, 'gelu': 2}[mode])
    return out


def add(a, b):
    """a + b (elementwise, same shape) on the q_sample kernel."""
    B = a.shape[0]
    one = torch.ones((B,), device=a.device, dtype=torch.float32)
    return q_sample(a.contiguous(), b.contiguous(), one, one)


def linear_backward(x, w, dy, want_bias=True):
    """y = x @ w.T (+ b), x (B,in), w (out,in), dy (B,out) -> dx (B,in), dw (out,in), db (out,)"""
    Bn, cin = x.shape
    cout = w.shape[0]
    x, dy, w = x.contiguous(), dy.contiguous(), w.contiguous()
    dx = _empty((Bn, cin), x)
    dw = _empty((cout, cin), x)
    if cout >= 1024 and cout % 128 == 0:       # long K (the stacked per-block MLPs): split it, add the parts in order
        nk = cout // 128
        part = _empty((nk, Bn, cin), x)
        bgemm(dy, (0, 128, cout, 1), w, (0, 128 * cin, cin, 1), part, (0, Bn * cin, cin, 1), Bn, cin, 128, 1, nk)
        call('dmh_sum_over_batch', ptr(part), ptr(dx), nk, Bn * cin)
    else:
        bgemm(dy, (0, 0, cout, 1), w, (0, 0, cin, 1), dx, (0, 0, cin, 1), Bn, cin, cout, 1, 1)
    bgemm(dy, (0, 0, 1, cout), x, (0, 0, cin, 1), dw, (0, 0, cin, 1), cout, cin, Bn, 1, 1)
    db = None
    if want_bias:
        db = _empty((cout,), x)
        call('dmh_sum_over_batch', ptr(dy), ptr(db), Bn, cout)
    return dx, dw, db


# ------------------------------------------------------------------ loss gradient + optimiser (training step)
def loss_backward(out, target, warped, mask, flow, abar, squared=False):
    """gradient of p_losses (CFG:796-806) wrt the UNet output (B,6,H,W); ``warped`` = flow_warp(out[:, 3:], flow)."""
    B, _, H, W = out.shape
    dout = torch.empty_like(out)
    gD = torch.empty((B, 3, H, W), device=out.device, dtype=F32)
    call('dmh_loss_backward', ptr(out), ptr(target.contiguous()), ptr(warped), ptr(mask.contiguous()),
         ptr(flow.contiguous()), ptr(abar.contiguous()), ptr(dout), ptr(gD), B, H, W, int(bool(squared)))
    return dout


def sumsq_blocks():
    return int(lib().dmh_sumsq_blocks())


def grad_norm_clip(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (DDP:1852) as data: (2,) tensor [total L2 norm, min(1, max_norm/(norm+1e-6))];
    the gradients themselves are left alone — dmh_adam multiplies by the coefficient while reading them."""
    nb = sumsq_blocks()
    dev = grads[0].device
    part = torch.empty((len(grads) * nb,), device=dev, dtype=torch.float64)
    for i, gr in enumerate(grads):
        call('dmh_sumsq', ptr(gr), gr.numel(), ptr(part[i * nb:], torch.float64))
    out = torch.empty((2,), device=dev, dtype=F32)
    call('dmh_gradnorm_finalize', ptr(part, torch.float64), part.numel(), float(max_norm), ptr(out))
    return out


def adam_(p, g, m, v, gscale, lr, b1, b2, eps, step):
    """in-place torch.optim.Adam update (no weight decay, no amsgrad) of p with gradient g * gscale[1]."""
    call('dmh_adam', ptr(p), ptr(g), ptr(m), ptr(v), ptr(gscale), p.numel(), float(lr), float(b1), float(b2),
         float(eps), int(step))


def ema_(ema, p, decay):
    """ema <- ema * decay + p * (1 - decay), in place."""
    call('dmh_ema', ptr(ema), ptr(p), p.numel(), float(decay))


def _ptr_table(tensors):
    for t in tensors:
        assert t.is_cuda and t.dtype == F32 and t.is_contiguous()
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def clip_adam_multi_(params, grads, ms, vs, max_norm, lr, b1, b2, eps, step):
    """clip_grad_norm_ + Adam.step (DDP:1852-1857) over lists of tensors in a handful of launches (dmh_sumsq_multi,
    dmh_gradnorm_finalize, dmh_adam_multi).  -> (2,) tensor [total norm, clip coefficient]"""
    n = len(params)
    sizes = (C.c_int64 * n)(*[p.numel() for p in params])
    gt = _ptr_table(grads)
    nb = int(lib().dmh_multi_blocks(sizes, n))
    part = torch.empty((nb,), device=params[0].device, dtype=torch.float64)
    call('dmh_sumsq_multi', gt, sizes, n, ptr(part, torch.float64))
    clip = torch.empty((2,), device=params[0].device, dtype=F32)
    call('dmh_gradnorm_finalize', ptr(part, torch.float64), nb, float(max_norm), ptr(clip))
    call('dmh_adam_multi', _ptr_table(params), gt, _ptr_table(ms), _ptr_table(vs), sizes, n, ptr(clip), float(lr), float(b1),
         float(b2), float(eps), int(step))
    return clip

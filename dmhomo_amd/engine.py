"""UNet execution on the gfx950 kernels: weight preparation + the launch program.

One ``UnetEngine`` serves both UNets of the reference (CFG = classifier_free_guidance.py,
DDP = denoising_diffusion_pytorch.py); which one is decided by the parameters present.
``prepare`` folds weight standardisation (N1), repacks every conv into the tile-major
layout of ``dmh_conv2d`` and transposes the small linears — once per weight version.
``forward_rows`` is the launch sequence of Unet.forward (CFG:412-466 / DDP:408-447):

  ResnetBlock  = conv3x3(+GN stats) -> gn_finalize(scale/shift) -> conv3x3 with fused
                 GN+SiLU prologue (+stats) -> gn_finalize -> SiLU(GN)+residual (fused into the
                 1x1 res_conv epilogue when the block changes width)
  LinearAttn   = pixel stats -> fused LN + to_qkv + context -> merge -> fused LN + to_q + apply -> 1x1 out ->
                 channel-LN + x   (q, k, v never stored; unfused fallback when C % 32 != 0)
  Attention    = channel-LN -> 1x1 qkv -> flash core -> 1x1 out (+x in the epilogue)
  torch.cat    never materialises: convs take two source pointers
  Upsample     nearest x2 is index math inside the 3x3 gather
"""
import math
import os

import torch

from . import ops

HEADS, DIM_HEAD = 4, 32     # CFG:246,275
ATTN_SCALE = DIM_HEAD ** -0.5
# development knob: '0' = final_conv as its own launch over the stored output of the last ResnetBlock
FUSED_FINAL = os.environ.get('DMH_FUSED_FINAL', '1') != '0'
FUSED_LINATTN = os.environ.get('DMH_FUSED_LINATTN', '1') != '0'   # development knob: '0' = separate LayerNorm / to_qkv / core


class _Res:
    __slots__ = ('conv1', 'g1', 'b1', 'conv2', 'g2', 'b2', 'res', 'ss_off', 'cout')


class _Attn:
    __slots__ = ('linear', 'ln_g', 'qkv', 'pla', 'plo', 'out', 'out_g')


def _ceil4(c):
    return (c + 3) // 4 * 4


class UnetEngine:
    def __init__(self, module, groups=8):
        self.module = module
        self.groups = groups
        self._sig = None
        self._rows = None      # the active-row list of the trunk pass being launched (ops.rows_from_keep), or None

    # ------------------------------------------------------------------ weights
    def _signature(self):
        # _dmh_epoch: bumped by train.TrainStep, whose kernels update the weights behind torch's version counters
        return (getattr(self.module, '_dmh_epoch', 0),) + tuple((p.data_ptr(), p._version) for p in self.module.parameters())

    def ensure_prepared(self):
        sig = self._signature()
        if sig != self._sig:
            self.prepare()
            self._sig = sig

    def prepare(self):
        sd = {k: v.detach() for k, v in self.module.named_parameters()}
        dev = next(iter(sd.values())).device
        if dev.type != 'cuda':
            raise RuntimeError('dmhomo_amd: the UNet must live on the GPU (call .cuda()); there is no CPU path')
        f32 = lambda k: sd[k].to(torch.float32).contiguous()

        def conv(key, c0, c1=0, ws=False, stride=1, ups=0, bias=True):
            w = f32(key + '.weight')
            if ws:
                w = ops.ws_standardize(w)
            b = f32(key + '.bias') if bias and (key + '.bias') in sd else None
            return ops.PackedConv(w, b, c0, c1, stride, ups)

        self.dim = sd['time_mlp.1.weight'].shape[1]
        self.has_classes = 'classes_emb.weight' in sd
        emb_dim = sd['time_mlp.3.weight'].shape[0] * (2 if self.has_classes else 1)
        self.emb_dim = emb_dim

        # N7 frequency table, computed as the reference does it on the host (CFG:167-169) — or the (learned / random) weights
        # of RandomOrLearnedSinusoidalPosEmb (CFG:175-190), whose embedding is (t, sin, cos): time_mlp.1 then takes dim + 1
        self.fourier_w = f32('time_mlp.0.weights') if 'time_mlp.0.weights' in sd else None
        if self.fourier_w is None:
            half = self.dim // 2
            f = math.log(10000) / (half - 1)
            self.freq = torch.exp(torch.arange(half) * -f).to(dev)
        self.t_w1, self.t_b1 = f32('time_mlp.1.weight').t().contiguous(), f32('time_mlp.1.bias')
        self.t_w2, self.t_b2 = f32('time_mlp.3.weight').t().contiguous(), f32('time_mlp.3.bias')
        if self.has_classes:
            self.c_table, self.c_null = f32('classes_emb.weight'), f32('null_classes_emb')
            self.c_w1, self.c_b1 = f32('classes_mlp.0.weight').t().contiguous(), f32('classes_mlp.0.bias')
            self.c_w2, self.c_b2 = f32('classes_mlp.2.weight').t().contiguous(), f32('classes_mlp.2.bias')

        # init conv: input channels padded to a multiple of 4 with zero weights
        w0 = f32('init_conv.weight')
        self.cin, self.cin_pad = w0.shape[1], _ceil4(w0.shape[1])
        if self.cin_pad != self.cin:
            wp = torch.zeros((w0.shape[0], self.cin_pad, 7, 7), device=dev, dtype=torch.float32)
            wp[:, :self.cin] = w0
            w0 = wp
        self.init_conv = ops.PackedConv(w0, f32('init_conv.bias'), self.cin_pad)

        mlp_w, mlp_b = [], []
        self._ss_total = 0

        def res(prefix, c0, c1=0):
            r = _Res()
            cout = sd[prefix + '.block1.proj.weight'].shape[0]
            r.cout = cout
            r.conv1 = conv(prefix + '.block1.proj', c0, c1, ws=True)
            r.g1, r.b1 = f32(prefix + '.block1.norm.weight'), f32(prefix + '.block1.norm.bias')
            r.conv2 = conv(prefix + '.block2.proj', cout, ws=True)
            r.g2, r.b2 = f32(prefix + '.block2.norm.weight'), f32(prefix + '.block2.norm.bias')
            r.res = conv(prefix + '.res_conv', c0, c1) if (prefix + '.res_conv.weight') in sd else None
            r.ss_off = self._ss_total
            mlp_w.append(f32(prefix + '.mlp.1.weight').t())
            mlp_b.append(f32(prefix + '.mlp.1.bias'))
            self._ss_total += 2 * cout
            return r

        def attn(prefix, c, linear):
            a = _Attn()
            a.linear = linear
            a.ln_g = f32(prefix + '.fn.norm.g').reshape(-1).contiguous()
            a.pla = None
            if linear and c % 32 == 0 and FUSED_LINATTN:
                # LayerNorm + to_qkv + both attention passes in two kernels (ops.linear_attention_fused)
                a.qkv = None
                a.pla = ops.PackedLinAttn(f32(prefix + '.fn.fn.to_qkv.weight'))
            else:
                a.qkv = conv(prefix + '.fn.fn.to_qkv', c, bias=False)
            a.plo = None
            if linear:
                a.out = conv(prefix + '.fn.fn.to_out.0', HEADS * DIM_HEAD)
                a.out_g = f32(prefix + '.fn.fn.to_out.1.g').reshape(-1).contiguous()
                if a.pla is not None and c == 64:   # to_out + LayerNorm + residual ride in the second fused pass
                    a.plo = ops.PackedLinAttnOut(f32(prefix + '.fn.fn.to_out.0.weight'),
                                                 f32(prefix + '.fn.fn.to_out.0.bias'), a.out_g)
            else:
                a.out = conv(prefix + '.fn.fn.to_out', HEADS * DIM_HEAD)
                a.out_g = None
            return a

        ns = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('downs.'))
        c = sd['init_conv.weight'].shape[0]
        self.init_dim = c
        self.downs, skip_c = [], []
        for i in range(ns):
            p = f'downs.{i}'
            b1 = res(p + '.0', c)
            skip_c.append(c)
            b2 = res(p + '.1', c)
            at = attn(p + '.2', c, True)
            skip_c.append(c)
            if (p + '.3.1.weight') in sd:                       # DDP: pixel-unshuffle + 1x1  == 2x2 / stride 2
                w = f32(p + '.3.1.weight')
                cout = w.shape[0]
                w22 = w.reshape(cout, c, 2, 2).contiguous()     # channel index = c*4 + p1*2 + p2 (DDP:112)
                down = ops.PackedConv(w22, f32(p + '.3.1.bias'), c, 0, stride=2)
            else:
                k = sd[p + '.3.weight'].shape[-1]
                down = conv(p + '.3', c, stride=2 if k == 4 else 1)
            self.downs.append((b1, b2, at, down))
            c = down.cout
        self.mid1 = res('mid_block1', c)
        self.mid_attn = attn('mid_attn', c, False)
        self.mid2 = res('mid_block2', c)
        self.ups = []
        for i in range(ns):
            p = f'ups.{i}'
            s1 = skip_c.pop()
            b1 = res(p + '.0', c, s1)
            c = b1.cout
            s2 = skip_c.pop()
            b2 = res(p + '.1', c, s2)
            c = b2.cout
            at = attn(p + '.2', c, True)
            if (p + '.3.1.weight') in sd:
                up = conv(p + '.3.1', c, ups=1)
            else:
                up = conv(p + '.3', c)
            self.ups.append((b1, b2, at, up))
            c = up.cout
        self.final_res = res('final_res_block', c, self.init_dim)
        self.final_w = f32('final_conv.weight').reshape(sd['final_conv.weight'].shape[0], -1).contiguous()
        self.final_b = f32('final_conv.bias')
        self.mlp_wt = torch.cat(mlp_w, dim=1).contiguous()       # (emb_dim, total)
        self.mlp_b = torch.cat(mlp_b).contiguous()

    # ------------------------------------------------------------------ blocks
    def _res(self, r, x0, x1, ss_all, pixel_stats=False, pre=None, final=None, keep_out=True, fin_out=None):
        """pixel_stats: return (x, stats) with the channel-LayerNorm statistics of x for the LinearAttention that
        follows (None where the block's last kernel cannot produce them).  pre: (y1, st1) = block1's convolution of x0
        and its GroupNorm partials, computed by the caller (``first_conv``).  final = (w, b): the block ends in its res_conv
        launch, which also applies the UNet's final 1x1 projection to every finished pixel -> (x or None, y NCHW)."""
        B, H, W, _ = x0.shape
        hw = H * W
        rows = self._rows
        y1, st1 = pre if pre is not None else ops.conv2d(r.conv1, x0, x1, want_stats=True, rows=rows)
        ss = ss_all[:, r.ss_off:r.ss_off + 2 * r.cout]
        if ops.STATIC_BOUND:
            coef1, bound1 = ops.gn_finalize(st1, r.g1, r.b1, hw, self.groups, ss, want_bound=True, rows=rows)
        else:
            coef1, bound1 = ops.gn_finalize(st1, r.g1, r.b1, hw, self.groups, ss, rows=rows), None
        y2, st2 = ops.conv2d(r.conv2, y1, in_coef=coef1, want_stats=True, in_bound=bound1, rows=rows)
        coef2 = ops.gn_finalize(st2, r.g2, r.b2, hw, self.groups, rows=rows)
        if final is not None:
            return ops.conv2d(r.res, x0, x1, res=y2, res_coef=coef2, final=final, keep_out=keep_out, fin_out=fin_out, rows=rows)
        if r.res is not None:
            if pixel_stats and r.cout == 64 and r.res.k == 1 and 64 in ops.PIXEL_STATS_FUSABLE and ops.f16x3_default() \
                    and os.environ.get('DMH_CONV_PIXEL_STATS', '1') != '0':      # (knob: same-box A/Bs)
                # (the up path at dim 64: the LayerNorm statistics of the LinearAttention behind this block come out of the
                #  res_conv launch that finishes it — DmhConv.pix_stats — instead of a dmh_pixel_stats pass over its output)
                return ops.conv2d(r.res, x0, x1, res=y2, res_coef=coef2, pixel_stats=True, rows=rows)
            x = ops.conv2d(r.res, x0, x1, res=y2, res_coef=coef2, rows=rows)
            return (x, None) if pixel_stats else x
        assert x1 is None
        if pixel_stats:
            if r.cout in ops.PIXEL_STATS_FUSABLE:
                return ops.gn_silu_residual(y2, coef2, x0, pixel_stats=True, rows=rows)
            return ops.gn_silu_residual(y2, coef2, x0, rows=rows), None
        return ops.gn_silu_residual(y2, coef2, x0, rows=rows)

    def _attn(self, a, x, stats=None):
        rows = self._rows
        if a.plo is not None:
            return ops.linear_attention_fused(x, a.ln_g, a.pla, ATTN_SCALE, out=a.plo, stats=stats, rows=rows)
        if a.pla is not None:
            o = ops.linear_attention_fused(x, a.ln_g, a.pla, ATTN_SCALE, stats=stats, rows=rows)
            y = ops.conv2d(a.out, o, rows=rows)
            return ops.chan_layernorm(y, a.out_g, res=x, rows=rows)
        xn = ops.chan_layernorm(x, a.ln_g, rows=rows)
        qkv = ops.conv2d(a.qkv, xn, rows=rows)
        if a.linear:
            o = ops.linear_attention_core(qkv, ATTN_SCALE, rows=rows)
            y = ops.conv2d(a.out, o, rows=rows)
            return ops.chan_layernorm(y, a.out_g, res=x, rows=rows)
        o = ops.attention_core(qkv, ATTN_SCALE, rows=rows)
        return ops.conv2d(a.out, o, res=x, rows=rows)

    # ------------------------------------------------------------------ program
    def embed(self, time, class_rows, reps):
        """cond rows (reps*B, emb_dim): [time_mlp(t) | classes_mlp(c)] (pre-SiLU), CFG:427,435,231-232."""
        B = time.shape[0]
        td = self.t_w2.shape[1]
        cond = torch.empty((reps * B, self.emb_dim), device=time.device, dtype=torch.float32)
        se = ops.fourier_embed(time, self.fourier_w) if self.fourier_w is not None else ops.sinusoidal_embed(time, self.freq)
        hmid = ops.linear(se, self.t_w1, self.t_b1, act_out='gelu')
        for r in range(reps):
            ops.linear(hmid, self.t_w2, self.t_b2, out=cond[r * B:(r + 1) * B, :td])
        if self.has_classes:
            for r, (classes, keep) in enumerate(class_rows):
                ce = ops.class_embed(classes, keep, self.c_table, self.c_null)
                cm = ops.linear(ce, self.c_w1, self.c_b1, act_out='gelu')
                ops.linear(cm, self.c_w2, self.c_b2, out=cond[r * B:(r + 1) * B, td:])
        return cond

    def ss_tables(self, times):
        """tables for ``ops.ss_gather`` (a replayed sampling loop): T (S, N) = the time half's contribution to the (scale, shift)
        linear for every timestep of ``times``, C (num_classes + 1, N) = the class half's for every class and (last row) the
        null embedding — each made by the SAME launches the per-step path uses (``embed``, then the linear over the
        concatenated mlp weight without its bias) on embedding rows whose other half is zero, so that (T + C) + bias is bit for
        bit the per-step result (the linear adds its input quarters as (q0 + q1) + (q2 + q3): time | class)."""
        self.ensure_prepared()
        assert self.has_classes and self.emb_dim == 2 * self.t_w2.shape[1] and self.emb_dim % 4 == 0
        dev, td = self.mlp_wt.device, self.t_w2.shape[1]
        t = torch.as_tensor(list(times), dtype=torch.int64).to(dev)
        ncls = self.c_table.shape[0]
        cls = torch.arange(ncls + 1, dtype=torch.int64, device=dev).clamp(max=ncls - 1)
        keep = torch.ones((ncls + 1,), dtype=torch.uint8, device=dev)
        keep[ncls] = 0
        cond_t = self.embed(t, [(torch.zeros_like(t), None)], 1)
        cond_t[:, td:].zero_()
        cond_c = self.embed(torch.zeros((ncls + 1,), dtype=torch.int64, device=dev), [(cls, keep)], 1)
        cond_c[:, :td].zero_()
        T = ops.linear(cond_t, self.mlp_wt, None, act_in='silu')
        Ct = ops.linear(cond_c, self.mlp_wt, None, act_in='silu')
        return T, Ct

    def stem(self, xin):
        """init_conv (CFG:432) on the assembled NHWC input; its output is shared by every CFG pass."""
        self.ensure_prepared()
        return ops.conv2d(self.init_conv, xin)

    def first_conv(self, x0):
        """block1's convolution of downs.0.0 on the stem output (+ its GroupNorm partials).  The embedding enters a
        ResnetBlock only behind that GroupNorm (scale / shift, CFG:206-210, 233-235), so this convolution — like the stem — is
        the same for every classifier-free-guidance pass of a sample and is computed once per sample (CFG:404 and :409 each
        run it; the rows are bitwise the same)."""
        self.ensure_prepared()
        return ops.conv2d(self.downs[0][0].conv1, x0, None, want_stats=True)

    def trunk(self, x0, cond, taps=None, first=None, out=None, ss_all=None, rows=None):
        """everything after init_conv.  x0: stem() output with one row per row of ``cond``.
        first: ``first_conv(x0)`` when the caller has it already (shared between the CFG passes).
        out: a contiguous (rows, out_dim, H, W) tensor that receives the result (a row slice of the caller's buffer).
        ss_all: the (scale, shift) rows when the caller has them already (``ss_tables`` + ``ops.ss_gather``); ``cond`` is then unused.
        rows: the active-row list of this pass (``ops.rows_from_keep``): every launch works on the listed rows of its tensors
        only — the other rows of every intermediate and of the result stay unwritten (Unet.dedup_dropped_rows).
        ``taps`` (dict) optionally receives the NHWC activation after each stage member, keyed like the
        reference's module names ('downs.0.0', 'mid_attn', ...): per-layer parity tests."""
        self.ensure_prepared()
        self._rows = rows
        try:
            return self._trunk(x0, cond, taps, first, out, ss_all)
        finally:
            self._rows = None

    def _trunk(self, x0, cond, taps, first, out, ss_all):
        rows = self._rows

        def tap(name, v):
            if taps is not None:
                taps[name] = v
            return v
        if ss_all is None:
            ss_all = ops.linear(cond, self.mlp_wt, self.mlp_b, act_in='silu')
        x = tap('init_conv', x0)
        r = x
        hs = []
        for i, (b1, b2, at, down) in enumerate(self.downs):
            x = tap(f'downs.{i}.0', self._res(b1, x, None, ss_all, pre=first if i == 0 else None))
            hs.append(x)
            x, pst = self._res(b2, x, None, ss_all, pixel_stats=True) if at.pla is not None else \
                (self._res(b2, x, None, ss_all), None)
            tap(f'downs.{i}.1', x)
            x = tap(f'downs.{i}.2', self._attn(at, x, pst))
            hs.append(x)
            x = tap(f'downs.{i}.3', ops.conv2d(down, x, rows=rows))
        x = tap('mid_block1', self._res(self.mid1, x, None, ss_all))
        x = tap('mid_attn', self._attn(self.mid_attn, x))
        x = tap('mid_block2', self._res(self.mid2, x, None, ss_all))
        for i, (b1, b2, at, up) in enumerate(self.ups):
            x = tap(f'ups.{i}.0', self._res(b1, x, hs.pop(), ss_all))
            x, pst = self._res(b2, x, hs.pop(), ss_all, pixel_stats=True) if at.pla is not None else \
                (self._res(b2, x, hs.pop(), ss_all), None)
            tap(f'ups.{i}.1', x)
            x = tap(f'ups.{i}.2', self._attn(at, x, pst))
            x = tap(f'ups.{i}.3', ops.conv2d(up, x, rows=rows))
        fr = self.final_res
        if FUSED_FINAL and fr.res is not None and fr.res.k == 1 and fr.cout <= 64 and self.final_w.shape[0] <= 8 \
                and ops.f16x3_default():
            # final_conv (CFG:341, 471-472) rides on the last block's res_conv launch: the block's output is projected
            # while it is still in registers, and only stored when a parity test taps it
            x, y = self._res(fr, x, r, ss_all, final=(self.final_w, self.final_b), keep_out=taps is not None, fin_out=out)
            tap('final_res_block', x)
            return y
        x = tap('final_res_block', self._res(fr, x, r, ss_all))
        y = ops.final_conv_nchw(x, self.final_w, self.final_b)
        if out is not None:
            out.copy_(y)
            return out
        return y

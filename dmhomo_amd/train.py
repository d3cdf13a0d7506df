"""Training step of the DGM diffusion model on the HIP kernels (SURVEY.md §8f "next" row 1).

``ResnetBlockTrain`` runs the forward of one ResnetBlock (CFG:216-241) on the same kernels as sampling while saving what
the backward pass needs, and the backward pass itself on the HIP kernels of csrc/conv_backward.hip and
csrc/norm_backward.hip:

  forward   x -> conv3x3(WS(w1)) -> GN(g1,b1)*(scale+1)+shift -> SiLU -> conv3x3(WS(w2)) -> GN(g2,b2) -> SiLU -> + res(x)
  saved     y1, y2 (conv outputs), the GroupNorm affines coef1/coef2 and (mean, rstd) mr1/mr2  (h = SiLU(GN(y1)) is
            never stored: consumers re-apply the affine + SiLU while staging, exactly as the forward conv does)
  backward  GN+SiLU backward (3 kernels) -> conv weight / bias gradient (fp32 MFMA, pixels as the K axis) -> weight
            standardisation backward -> data gradient = the forward conv kernel with the flipped, transposed weight

``UnetTrain`` strings those blocks, the attention / LayerNorm / embedding / resampling-conv backward kernels into the
whole conditional UNet (CFG:412-466) with a manual tape; ``TrainStep`` adds p_losses (CFG:770-806) with its gradient,
gradient accumulation, the global-norm clip, Adam and the RCCL gradient all-reduce: one optimiser step of
``Trainer.train`` (DDP:1830-1862).  There is no autograd anywhere on this path.
"""
import os

import torch

from . import ops


class ResnetBlockTrain:
    """one ResnetBlock with raw (trainable) parameters as device tensors.  ``p`` keys: w1 b1 g1 be1 w2 b2 g2 be2 and,
    when the block changes width, rw rb (the 1x1 res_conv).  c1 > 0: the input is cat(x0, x1) (up path)."""

    def __init__(self, p, c0, c1=0, groups=8, batch=None):
        self.p, self.c0, self.c1, self.groups = p, c0, c1, groups
        self.build(batch)

    def build(self, batch=None):
        """standardised weights, forward and data-gradient images.  With ``batch`` (ops.PackBatch) the buffers are only
        allocated and registered here: ``batch.run()`` fills them, now and after every parameter update."""
        p = self.p
        cout = p['w1'].shape[0]
        self.cout = cout
        if batch is not None and ops._batchable(3, 1, 0):
            self.w1s, self.w2s = torch.empty_like(p['w1']), torch.empty_like(p['w2'])
            self.f1 = ops.PackedConv(self.w1s, p['b1'], self.c0, self.c1, batch=batch, ws_from=p['w1'])
            self.f2 = ops.PackedConv(self.w2s, p['b2'], cout, batch=batch, ws_from=p['w2'])
        elif batch is not None:
            # DMH_CONV3_VARIANT selects the exact-fp32 kernels: dmh_pack_conv_weights_multi does not make their images, so
            # the standardised weights are written into fixed buffers ahead of the per-weight pack launches
            self.w1s, self.w2s = ops.ws_standardize(p['w1']), ops.ws_standardize(p['w2'])
            batch.pre.append(lambda: (self.w1s.copy_(ops.ws_standardize(self.p['w1'])),
                                      self.w2s.copy_(ops.ws_standardize(self.p['w2']))))
            self.f1 = ops.PackedConv(self.w1s, p['b1'], self.c0, self.c1, batch=batch)
            self.f2 = ops.PackedConv(self.w2s, p['b2'], cout, batch=batch)
        else:
            self.w1s, self.w2s = ops.ws_standardize(p['w1']), ops.ws_standardize(p['w2'])
            self.f1 = ops.PackedConv(self.w1s, p['b1'], self.c0, self.c1)
            self.f2 = ops.PackedConv(self.w2s, p['b2'], cout)
        self.d1 = ops.conv_dgrad_pack(self.w1s, self.c0 + self.c1, batch=batch)
        self.d2 = ops.conv_dgrad_pack(self.w2s, cout, batch=batch)
        self.fr = self.dr = None
        if 'rw' in p:
            self.fr = ops.PackedConv(p['rw'], p['rb'], self.c0, self.c1, batch=batch)
            self.dr = ops.conv_dgrad_pack(p['rw'], self.c0 + self.c1, batch=batch)

    def forward(self, x0, x1=None, ss=None):
        """x0 (B,H,W,c0) [, x1 (B,H,W,c1)], ss (B, 2*cout) = (scale, shift) of the mlp -> out (B,H,W,cout), saved"""
        p, hw = self.p, x0.shape[1] * x0.shape[2]
        y1, st1 = ops.conv2d(self.f1, x0, x1, want_stats=True)
        coef1, mr1 = ops.gn_finalize_train(st1, p['g1'], p['be1'], hw, self.groups, ss)
        y2, st2 = ops.conv2d(self.f2, y1, in_coef=coef1, want_stats=True)
        coef2, mr2 = ops.gn_finalize_train(st2, p['g2'], p['be2'], hw, self.groups)
        if self.fr is not None:
            out = ops.conv2d(self.fr, x0, x1, res=y2, res_coef=coef2)
        else:
            out = ops.gn_silu_residual(y2, coef2, x0)
        return out, dict(x0=x0, x1=x1, ss=ss, y1=y1, coef1=coef1, mr1=mr1, y2=y2, coef2=coef2, mr2=mr2)

    def backward(self, sv, dout):
        """dout (B,H,W,cout) -> dx (B,H,W,c0+c1) (the caller slices the two concat halves), grads {name: tensor}"""
        p, g = self.p, {}
        dout = dout.contiguous()
        # out = SiLU(GN2(y2)) + res
        dy2, g['g2'], g['be2'], _ = ops.gn_silu_backward(dout, sv['y2'], sv['coef2'], sv['mr2'], p['g2'], p['be2'],
                                                         self.groups)
        dw2s, g['b2'] = ops.conv_wgrad(dy2, sv['y1'], k=3, in_coef=sv['coef1'])
        g['w2'] = ops.ws_backward(p['w2'], dw2s)
        dh1 = ops.conv2d(self.d2, dy2)                                    # gradient wrt h1 = SiLU(GN1(y1)...)
        dy1, g['g1'], g['be1'], g['ss'] = ops.gn_silu_backward(dh1, sv['y1'], sv['coef1'], sv['mr1'], p['g1'], p['be1'],
                                                               self.groups, ss=sv['ss'])
        dw1s, g['b1'] = ops.conv_wgrad(dy1, sv['x0'], sv['x1'], k=3)
        g['w1'] = ops.ws_backward(p['w1'], dw1s)
        if self.fr is not None:
            g['rw'], g['rb'] = ops.conv_wgrad(dout, sv['x0'], sv['x1'], k=1)
            dres = ops.conv2d(self.dr, dout)                              # through the 1x1 res_conv
        else:
            dres = dout                                                   # identity residual
        dx = ops.conv2d(self.d1, dy1, res=dres)                           # data gradient of conv1 + the residual path
        return dx, g


# =====================================================================================================================
# the whole conditional UNet (CFG:302-466): forward with saved activations + backward, on the HIP kernels
# =====================================================================================================================
HEADS, DIM_HEAD = 4, 32
SCALE = DIM_HEAD ** -0.5


class _LinAttn:
    """Residual(PreNorm(LinearAttention)) (CFG:96-103, 246-269) or, with linear=False, Residual(PreNorm(Attention))
    (CFG:273-296), on a stored qkv tensor."""

    def __init__(self, p, c, linear, batch=None):
        self.p, self.c, self.linear = p, c, linear
        self.fq = ops.PackedConv(p['qkv'], None, c, batch=batch)
        self.dq = ops.conv_dgrad_pack(p['qkv'], c, batch=batch)
        self.fo = ops.PackedConv(p['ow'], p['ob'], HEADS * DIM_HEAD, batch=batch)
        self.do = ops.conv_dgrad_pack(p['ow'], HEADS * DIM_HEAD, batch=batch)

    def forward(self, x):
        p = self.p
        xn = ops.chan_layernorm(x, p['g'])
        qkv = ops.conv2d(self.fq, xn)
        if self.linear:
            o, core = ops.linear_attention_core_train(qkv, SCALE)
            y = ops.conv2d(self.fo, o)
            out = ops.chan_layernorm(y, p['og'], res=x)
        else:
            o, core = ops.attention_core_train(qkv, SCALE)
            y = None
            out = ops.conv2d(self.fo, o, res=x)
        return out, dict(x=x, xn=xn, o=o, y=y, core=core)

    def backward(self, sv, dout):
        p, g = self.p, {}
        dout = dout.contiguous()
        if self.linear:
            dy, g['og'] = ops.chan_layernorm_backward(sv['y'], p['og'], dout)
        else:
            dy = dout
        g['ow'], g['ob'] = ops.conv_wgrad(dy, sv['o'], k=1)
        do = ops.conv2d(self.do, dy)
        dqkv = (ops.linear_attention_core_backward if self.linear else ops.attention_core_backward)(sv['core'], do)
        g['qkv'] = ops.conv_wgrad(dqkv, sv['xn'], k=1, want_bias=False)
        dxn = ops.conv2d(self.dq, dqkv)
        dx, g['g'] = ops.chan_layernorm_backward(sv['x'], p['g'], dxn)
        return ops.add(dx, dout), g                                        # + the residual path


class _Conv:
    """plain biased conv of the trunk: kind 'same3' (3x3), 'down4' (4x4 / stride 2), 'up3' (nearest x2 + 3x3), 'init7'."""

    def __init__(self, w, b, kind, c, batch=None):
        self.w, self.b, self.kind, self.c = w, b, kind, c
        k = w.shape[-1]
        self.f = ops.PackedConv(w, b, c, 0, 2 if kind == 'down4' else 1, 1 if kind == 'up3' else 0, batch=batch)
        self.d = None
        if kind in ('same3', 'up3'):
            self.d = ops.conv_dgrad_pack(w, c, batch=batch)
        elif kind == 'down4':
            self.d = ops.conv_down_dgrad_pack(w, batch=batch)
        self.k = k

    def forward(self, x):
        return ops.conv2d(self.f, x), x

    def backward(self, x, dy, want_dx=True):
        if self.kind == 'down4':
            return ops.conv_down_backward(dy, x, self.w, self.d)
        if self.kind == 'up3':
            return ops.conv_up_backward(dy, x, self.w, self.d)
        dw, db = ops.conv_wgrad(dy, x, k=self.k)
        dx = ops.conv2d(self.d, dy) if (want_dx and self.d is not None) else None
        return dx, dw, db


class UnetTrain:
    """forward (saving activations) and backward of classifier_free_guidance.Unet on the HIP kernels.
    ``module``: a dmhomo_amd.cfg.Unet (same parameter names as the reference).  Gradients come back as
    {parameter name: tensor of the parameter's shape}."""

    def __init__(self, module, groups=8):
        self.module, self.groups = module, groups
        self.refresh()

    def repack(self):
        """the weight images from the parameters' CURRENT values (same buffers, same parameter storage as at the last
        ``refresh()``): three table-driven launches for the ~140 plain convolution weights + the few special ones; a
        handful of small tensors derived from parameters by torch ops are re-derived.  Graph-capturable."""
        self.pack.run()
        sd = self.sd
        self.mlp_w.copy_(torch.cat([sd[k + '.mlp.1.weight'] for k in self.mlp_names], 0))
        self.mlp_b.copy_(torch.cat([sd[k + '.mlp.1.bias'] for k in self.mlp_names], 0))

    def refresh(self):
        """build everything from the module's parameters (first use, or after their storage moved)"""
        sd = {k: v.detach().to(torch.float32).contiguous() for k, v in self.module.named_parameters()}
        self.sd = sd
        if 'time_mlp.0.weights' in sd or any(k.startswith('time_mlp.0.') for k, _ in self.module.named_buffers()):
            # RandomOrLearnedSinusoidalPosEmb (CFG:175-190): its embedding is learned_dim + 1 wide and time_mlp.1 is sized for
            # THAT — the sinusoidal embedding below would feed it a wrong K silently.  No training entry point reaches such a
            # model (GaussianDiffusion refuses it, CFG:514-515); anything else that does gets told.
            raise NotImplementedError('training a Unet with learned_sinusoidal_cond / random_fourier_features is not supported: '
                                      'the training step has no gradient for RandomOrLearnedSinusoidalPosEmb.weights')
        self.pack = pb = ops.PackBatch()
        self.dim = sd['time_mlp.1.weight'].shape[1]
        half = self.dim // 2
        import math
        if getattr(self, 'freq', None) is None:      # constant: built once (a host->device copy cannot sit in a HIP graph)
            self.freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(sd['init_conv.weight'].device)
        dev = sd['init_conv.weight'].device
        w0 = sd['init_conv.weight']
        self.cin = w0.shape[1]
        self.cin_pad = (self.cin + 3) // 4 * 4
        wp = torch.zeros((w0.shape[0], self.cin_pad, 7, 7), device=dev)
        cin = self.cin

        def pad_init():
            wp[:, :cin] = w0
        pad_init()
        pb.pre.append(pad_init)
        self.init = _Conv(wp, sd['init_conv.bias'], 'init7', self.cin_pad, batch=pb)
        self.blocks, self.ss_off, self.ss_total = {}, {}, 0

        def res(prefix, c0, c1=0):
            p = dict(w1=sd[prefix + '.block1.proj.weight'], b1=sd[prefix + '.block1.proj.bias'],
                     g1=sd[prefix + '.block1.norm.weight'], be1=sd[prefix + '.block1.norm.bias'],
                     w2=sd[prefix + '.block2.proj.weight'], b2=sd[prefix + '.block2.proj.bias'],
                     g2=sd[prefix + '.block2.norm.weight'], be2=sd[prefix + '.block2.norm.bias'])
            if (prefix + '.res_conv.weight') in sd:
                p['rw'], p['rb'] = sd[prefix + '.res_conv.weight'], sd[prefix + '.res_conv.bias']
            blk = ResnetBlockTrain(p, c0, c1, self.groups, batch=pb)
            self.blocks[prefix] = blk
            self.ss_off[prefix] = self.ss_total
            self.ss_total += 2 * blk.cout
            return blk

        def attn(prefix, c, linear):
            p = dict(g=sd[prefix + '.fn.norm.g'].reshape(-1).contiguous(), qkv=sd[prefix + '.fn.fn.to_qkv.weight'])
            if linear:
                p['ow'], p['ob'] = sd[prefix + '.fn.fn.to_out.0.weight'], sd[prefix + '.fn.fn.to_out.0.bias']
                p['og'] = sd[prefix + '.fn.fn.to_out.1.g'].reshape(-1).contiguous()
            else:
                p['ow'], p['ob'] = sd[prefix + '.fn.fn.to_out.weight'], sd[prefix + '.fn.fn.to_out.bias']
            a = _LinAttn(p, c, linear, batch=pb)
            self.blocks[prefix] = a
            return a

        ns = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('downs.'))
        self.ns = ns
        c = w0.shape[0]
        self.init_dim = c
        skip_c = []
        for i in range(ns):
            pfx = f'downs.{i}'
            b1 = res(pfx + '.0', c)
            skip_c.append(c)
            res(pfx + '.1', c)
            attn(pfx + '.2', c, True)
            skip_c.append(c)
            w = sd[pfx + '.3.weight']
            self.blocks[pfx + '.3'] = _Conv(w, sd[pfx + '.3.bias'], 'down4' if w.shape[-1] == 4 else 'same3', c, batch=pb)
            c = w.shape[0]
        res('mid_block1', c)
        attn('mid_attn', c, False)
        res('mid_block2', c)
        for i in range(ns):
            pfx = f'ups.{i}'
            b1 = res(pfx + '.0', c, skip_c.pop())
            c = b1.cout
            b2 = res(pfx + '.1', c, skip_c.pop())
            c = b2.cout
            attn(pfx + '.2', c, True)
            if (pfx + '.3.1.weight') in sd:
                w = sd[pfx + '.3.1.weight']
                self.blocks[pfx + '.3'] = _Conv(w, sd[pfx + '.3.1.bias'], 'up3', c, batch=pb)
            else:
                w = sd[pfx + '.3.weight']
                self.blocks[pfx + '.3'] = _Conv(w, sd[pfx + '.3.bias'], 'same3', c, batch=pb)
            c = w.shape[0]
        res('final_res_block', c, self.init_dim)
        self.mlp_names = [k for k in self.ss_off]          # creation order == offsets order
        self.mlp_w = torch.cat([sd[k + '.mlp.1.weight'] for k in self.mlp_names], 0).contiguous()      # (total, emb)
        self.mlp_b = torch.cat([sd[k + '.mlp.1.bias'] for k in self.mlp_names], 0).contiguous()
        self.final_w = sd['final_conv.weight'].reshape(sd['final_conv.weight'].shape[0], -1).contiguous()   # (6, 64)
        assert self.final_w.data_ptr() == sd['final_conv.weight'].data_ptr()      # a view: follows the parameter
        pb.run()                                                 # the first fill of every image registered above

    # ------------------------------------------------------------------ small dense layers
    @staticmethod
    def _lin(x, w, b):
        return ops.linear(x, w.t().contiguous(), b)

    def _embed_forward(self, time, classes, keep):
        sd = self.sd
        se = ops.sinusoidal_embed(time, self.freq)
        h1 = self._lin(se, sd['time_mlp.1.weight'], sd['time_mlp.1.bias'])
        a1 = ops.act(h1, 'gelu')
        temb = self._lin(a1, sd['time_mlp.3.weight'], sd['time_mlp.3.bias'])
        ce = ops.class_embed(classes, keep, sd['classes_emb.weight'], sd['null_classes_emb'])
        h2 = self._lin(ce, sd['classes_mlp.0.weight'], sd['classes_mlp.0.bias'])
        a2 = ops.act(h2, 'gelu')
        cemb = self._lin(a2, sd['classes_mlp.2.weight'], sd['classes_mlp.2.bias'])
        cond = torch.cat([temb, cemb], dim=1).contiguous()
        ac = ops.act(cond, 'silu')
        ss_all = self._lin(ac, self.mlp_w, self.mlp_b)
        return ss_all, dict(se=se, h1=h1, a1=a1, ce=ce, h2=h2, a2=a2, cond=cond, ac=ac, classes=classes, keep=keep)

    def _embed_backward(self, sv, dss_all, g):
        sd = self.sd
        dac, dw, db = ops.linear_backward(sv['ac'], self.mlp_w, dss_all)
        off = 0
        for k in self.mlp_names:
            n2 = 2 * self.blocks[k].cout
            g[k + '.mlp.1.weight'], g[k + '.mlp.1.bias'] = dw[off:off + n2].contiguous(), db[off:off + n2].contiguous()
            off += n2
        dcond = ops.act(sv['cond'], 'silu', dy=dac)
        td = sd['time_mlp.3.weight'].shape[0]
        dtemb, dcemb = dcond[:, :td].contiguous(), dcond[:, td:].contiguous()
        da1, g['time_mlp.3.weight'], g['time_mlp.3.bias'] = ops.linear_backward(sv['a1'], sd['time_mlp.3.weight'], dtemb)
        dh1 = ops.act(sv['h1'], 'gelu', dy=da1)
        _, g['time_mlp.1.weight'], g['time_mlp.1.bias'] = ops.linear_backward(sv['se'], sd['time_mlp.1.weight'], dh1)
        da2, g['classes_mlp.2.weight'], g['classes_mlp.2.bias'] = ops.linear_backward(sv['a2'], sd['classes_mlp.2.weight'], dcemb)
        dh2 = ops.act(sv['h2'], 'gelu', dy=da2)
        dce, g['classes_mlp.0.weight'], g['classes_mlp.0.bias'] = ops.linear_backward(sv['ce'], sd['classes_mlp.0.weight'], dh2)
        tab = sd['classes_emb.weight']
        dtab, dnull = torch.empty_like(tab), torch.empty_like(sd['null_classes_emb'])
        ops.call('dmh_class_embed_backward', ops.ptr(dce), ops.ptr(sv['classes'], torch.int64), ops.ptr(sv['keep'], torch.uint8),
                 ops.ptr(dtab), ops.ptr(dnull), dce.shape[0], dce.shape[1], tab.shape[0])
        g['classes_emb.weight'], g['null_classes_emb'] = dtab, dnull

    # ------------------------------------------------------------------ trunk
    def forward(self, x, time, classes, rgb_flow, mask, keep, taps=None):
        """x (B,6,H,W), rgb_flow (B,3,H,W), mask (B,1,H,W) NCHW fp32, time/classes (B,) int64, keep (B,) bool
        -> out (B,6,H,W) NCHW, saved"""
        B = x.shape[0]
        keep = keep.to(torch.uint8).contiguous()
        ss_all, emb = self._embed_forward(time, classes, keep)
        xin = ops.assemble_input(x.contiguous(), rgb_flow.contiguous(), mask.contiguous(), cpad=self.cin_pad)
        T = []                                                   # tape: (kind, block name, saved)

        def ss(prefix):
            o = self.ss_off[prefix]
            return ss_all[:, o:o + 2 * self.blocks[prefix].cout]

        def run_res(prefix, x0, x1=None):
            out, sv = self.blocks[prefix].forward(x0, x1, ss(prefix))
            T.append(('res', prefix, sv))
            if taps is not None:
                taps[prefix] = out
            return out

        def run_attn(prefix, x0):
            out, sv = self.blocks[prefix].forward(x0)
            T.append(('attn', prefix, sv))
            if taps is not None:
                taps[prefix] = out
            return out

        def run_conv(prefix, x0):
            out, sv = self.blocks[prefix].forward(x0)
            T.append(('conv', prefix, sv))
            if taps is not None:
                taps[prefix] = out
            return out

        h, _ = self.init.forward(xin)
        if taps is not None:
            taps['init_conv'] = h
            taps['ss_all'] = ss_all
        r = h
        hs = []
        for i in range(self.ns):
            h = run_res(f'downs.{i}.0', h)
            hs.append(h)
            h = run_res(f'downs.{i}.1', h)
            h = run_attn(f'downs.{i}.2', h)
            hs.append(h)
            h = run_conv(f'downs.{i}.3', h)
        h = run_res('mid_block1', h)
        h = run_attn('mid_attn', h)
        h = run_res('mid_block2', h)
        for i in range(self.ns):
            h = run_res(f'ups.{i}.0', h, hs.pop())
            h = run_res(f'ups.{i}.1', h, hs.pop())
            h = run_attn(f'ups.{i}.2', h)
            h = run_conv(f'ups.{i}.3', h)
        h = run_res('final_res_block', h, r)
        out = ops.final_conv_nchw(h, self.final_w, self.sd['final_conv.bias'])
        return out, dict(tape=T, emb=emb, xin=xin, hfinal=h, B=B)

    def backward(self, sv, dout):
        """dout (B,6,H,W) NCHW -> {parameter name: gradient}"""
        g = {}
        sd, T = self.sd, sv['tape']
        B = sv['B']
        h = sv['hfinal']
        _, H, W, cf = h.shape
        hw = H * W
        dout = dout.contiguous()
        no = dout.shape[1]
        # ---- final_conv (1x1, NCHW output): dh[p][c] = sum_o dout[o][p] w[o][c];  dw[o][c] = sum_{b,p} dout[o][p] h[p][c]
        dh = torch.empty_like(h)
        ops.bgemm(dout, (no * hw, 0, 1, hw), self.final_w, (0, 0, cf, 1), dh, (hw * cf, 0, cf, 1), hw, cf, no, B, 1)
        # (the pixel axis is the K of these two: split it over the inner batch index so the launch has waves to spare)
        ks = 256 if hw % 256 == 0 else hw
        nk = hw // ks
        dwb = ops._empty((B, nk, no, cf), h)
        ops.bgemm(dout, (no * hw, ks, hw, 1), h, (hw * cf, ks * cf, cf, 1), dwb, (nk * no * cf, no * cf, cf, 1), no, cf, ks,
                  B, nk)
        dw = ops._empty((no, cf), h)
        ops.call('dmh_sum_over_batch', ops.ptr(dwb), ops.ptr(dw), B * nk, no * cf)
        g['final_conv.weight'] = dw.reshape(sd['final_conv.weight'].shape)
        ones = torch.ones((ks, 1), device=h.device, dtype=torch.float32)
        dbb = ops._empty((B, nk, no), h)
        ops.bgemm(dout, (no * hw, ks, hw, 1), ones, (0, 0, 1, 1), dbb, (nk * no, no, 1, 1), no, 1, ks, B, nk)
        db = ops._empty((no,), h)
        ops.call('dmh_sum_over_batch', ops.ptr(dbb), ops.ptr(db), B * nk, no)
        g['final_conv.bias'] = db
        # ---- trunk, in reverse
        dss_all = torch.zeros((B, self.ss_total), device=h.device, dtype=torch.float32)
        names = {'w1': '.block1.proj.weight', 'b1': '.block1.proj.bias', 'g1': '.block1.norm.weight',
                 'be1': '.block1.norm.bias', 'w2': '.block2.proj.weight', 'b2': '.block2.proj.bias',
                 'g2': '.block2.norm.weight', 'be2': '.block2.norm.bias', 'rw': '.res_conv.weight', 'rb': '.res_conv.bias'}
        idx = len(T)

        def pop(kind):
            nonlocal idx
            idx -= 1
            k, prefix, s = T[idx]
            assert k == kind, (k, kind)
            return prefix, s

        def back_res(d):
            prefix, s = pop('res')
            blk = self.blocks[prefix]
            dx, gg = blk.backward(s, d)
            for k, v in gg.items():
                if k == 'ss':
                    o = self.ss_off[prefix]
                    dss_all[:, o:o + 2 * blk.cout] = v
                else:
                    g[prefix + names[k]] = v.reshape(sd[prefix + names[k]].shape)
            if blk.c1:
                # (the skip half stays a channel-slice VIEW: its one consumer is the strided in-place add below)
                return dx[..., :blk.c0].contiguous(), dx[..., blk.c0:]
            return dx, None

        def back_attn(d):
            prefix, s = pop('attn')
            a = self.blocks[prefix]
            dx, gg = a.backward(s, d)
            g[prefix + '.fn.norm.g'] = gg['g'].reshape(sd[prefix + '.fn.norm.g'].shape)
            g[prefix + '.fn.fn.to_qkv.weight'] = gg['qkv']
            if a.linear:
                g[prefix + '.fn.fn.to_out.0.weight'], g[prefix + '.fn.fn.to_out.0.bias'] = gg['ow'], gg['ob']
                g[prefix + '.fn.fn.to_out.1.g'] = gg['og'].reshape(sd[prefix + '.fn.fn.to_out.1.g'].shape)
            else:
                g[prefix + '.fn.fn.to_out.weight'], g[prefix + '.fn.fn.to_out.bias'] = gg['ow'], gg['ob']
            return dx

        def back_conv(d):
            prefix, x0 = pop('conv')
            cv = self.blocks[prefix]
            dx, dw_, db_ = cv.backward(x0, d)
            wname = prefix + ('.1.weight' if cv.kind == 'up3' else '.weight')
            g[wname], g[wname.replace('weight', 'bias')] = dw_, db_
            return dx

        d, dr = back_res(dh)                                     # final_res_block: input cat(x, r)
        skip_grads = []                                          # gradients of the skip tensors, in pop order
        for i in reversed(range(self.ns)):
            d = back_conv(d)
            d = back_attn(d)
            d, ds2 = back_res(d)                                 # ups.i.1: cat(x, hs.pop())
            d, ds1 = back_res(d)                                 # ups.i.0
            skip_grads.append((ds1, ds2))
        d, _ = back_res(d)                                       # mid_block2
        d = back_attn(d)
        d, _ = back_res(d)                                       # mid_block1
        # forward pushed [downs.0: a, b, downs.1: a, b, ...]; ups.0 popped downs.(ns-1).b then .a, ...
        for i in reversed(range(self.ns)):
            ds1, ds2 = skip_grads[i]      # appended for ups.(ns-1), ..., ups.0; ups.j pops the pushes of downs.(ns-1-j)
            d = back_conv(d)                                     # downs.i.3
            d.add_(ds1)                                          # downs.i.2's output was also pushed to the skip stack
            d = back_attn(d)
            d, _ = back_res(d)                                   # downs.i.1
            d.add_(ds2)                                          # downs.i.0's output was pushed too
            d, _ = back_res(d)                                   # downs.i.0
        d.add_(dr)                                               # r = init_conv output, also fed to final_res_block
        _, dw0, db0 = self.init.backward(sv['xin'], d, want_dx=False)
        g['init_conv.weight'], g['init_conv.bias'] = dw0[:, :self.cin].contiguous(), db0
        assert idx == 0
        self._embed_backward(sv['emb'], dss_all, g)
        return g


class TrainStep:
    """one optimiser step of the reference's training loop (DDP:1830-1862) for a dmhomo_amd.cfg.GaussianDiffusion:

        for _ in range(accum):  loss = diffusion(batch, classes=...) / accum;  loss.backward()      DDP:1839-1850
        clip_grad_norm_(parameters, 1.0);  opt.step();  opt.zero_grad()                             DDP:1852-1862

    with torch.optim.Adam(lr, betas) semantics (DDP:1741).  Gradients of ranks > 1 are averaged with one RCCL
    all-reduce over a flat buffer (what accelerate's DDP wrapper does for the reference)."""

    def __init__(self, diffusion, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, max_grad_norm=1.0, accum=1, groups=8):
        self.diffusion, self.unet = diffusion, diffusion.model
        self.lr, self.betas, self.eps = lr, betas, eps
        self.max_grad_norm, self.accum = max_grad_norm, accum
        self.params = dict(self.unet.named_parameters())
        for k, p in self.params.items():
            assert p.dtype == torch.float32 and p.is_contiguous(), k
        self.ut = UnetTrain(self.unet, groups)
        self._m = self._v = None          # Adam moments: allocated by the first optimiser step (apply / load_state_dict)
        self._sig = self._signature()
        self.opt_step = 0
        self.last_norm = None
        # the weight re-pack after every optimiser step (~1200 small launches: standardise, scale, split, pack for the
        # forward and the data-gradient convs) is captured once into a HIP graph and replayed: same buffers, one launch
        self._repack_graph = None if os.environ.get('DMH_TRAIN_GRAPH', '1') != '0' else False

    @property
    def m(self):
        if self._m is None:
            self._m = {k: torch.zeros_like(p) for k, p in self.params.items()}
        return self._m

    @property
    def v(self):
        if self._v is None:
            self._v = {k: torch.zeros_like(p) for k, p in self.params.items()}
        return self._v

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.params.values())

    def ensure_fresh(self):
        """re-pack when something other than ``apply`` changed the parameters (an external optimiser, load_state_dict)"""
        self.params = dict(self.unet.named_parameters())         # a re-materialised Parameter is a new object
        sig = self._signature()
        if sig != self._sig:
            if tuple(a for a, _ in sig) != tuple(a for a, _ in self._sig):
                # storage moved (diffusion.to() / .float(), a cloned or re-created parameter): the captured re-pack graph
                # reads the OLD pointers — drop it (re-captured on the next refresh); Adam moments follow the parameters
                if self._repack_graph is not False:
                    self._repack_graph = None
                for st in (self._m, self._v):
                    if st is not None:
                        for k, p in self.params.items():
                            if st[k].device != p.device:
                                st[k] = st[k].to(p.device)
            self.unet._dmh_epoch = getattr(self.unet, '_dmh_epoch', 0) + 1
            self.refresh()
            self._sig = sig

    # ---- forward + backward of GaussianDiffusion.forward (CFG:808-842) on one 12-channel batch
    def loss_and_grads(self, img, classes, t=None, noise=None, keep=None, grad_scale=1.0):
        """-> (loss (0-dim tensor, unscaled), {parameter name: d(grad_scale * loss)/d parameter}).  ``t``, ``noise``,
        ``keep`` default to the reference's draws, in its order (CFG:812 randint, CFG:773 randn_like, CFG:422 uniform)."""
        from .ddpm import flow_warp
        df = self.diffusion
        b, c, h, w = img.shape
        assert h == df.image_size and w == df.image_size, f'height and width of image must be {df.image_size}'
        self.ensure_fresh()
        dev = img.device
        if t is None:
            t = torch.randint(0, df.num_timesteps, (b,), device=dev).long()
        x_start = ops.affine(img[:, :6].to(torch.float32), 2., -1.)
        mask = img[:, 6:7].to(torch.float32).contiguous()
        rgb_flow = ops.affine(img[:, -5:-2].to(torch.float32), 2., -1.)
        flow = img[:, -2:].to(torch.float32).contiguous()
        if noise is None:
            noise = df.rng.randn(x_start.shape, dev)
        noise = noise.to(torch.float32).contiguous()
        t = t.to(torch.int64).contiguous()
        if keep is None:
            keep = self.unet._keep_mask(b, self.unet.cond_drop_prob, dev)
            if keep is None:
                keep = torch.ones((b,), device=dev, dtype=torch.uint8)
        squared = df.loss_fn == 'l2'
        x = df.q_sample(x_start, t, noise)
        out, saved = self.ut.forward(x, t, classes.to(torch.int64).contiguous(), rgb_flow, mask, keep)
        warped = flow_warp(out[:, 3:].contiguous(), flow)
        if df.objective == 'pred_noise':
            target = noise
        elif df.objective == 'pred_x0':
            target = x_start
        elif df.objective == 'pred_v':
            ca = df.sqrt_alphas_cumprod.gather(-1, t).contiguous()
            cb = (-df.sqrt_one_minus_alphas_cumprod).gather(-1, t).contiguous()
            target = ops.q_sample(noise, x_start, ca, cb)
        else:
            raise ValueError(f'unknown objective {df.objective}')
        abar = df.alphas_cumprod.gather(-1, t).to(torch.float32).contiguous()
        loss = ops.loss_combine(ops.diff_mean(out, target, None, squared),
                                ops.diff_mean(warped, out[:, :3].contiguous(), mask, squared), abar)
        dout = ops.loss_backward(out, target, warped, mask, flow, abar, squared)
        if grad_scale != 1.0:
            dout = ops.affine(dout, grad_scale, 0.)
        return loss, self.ut.backward(saved, dout)

    # ---- torch.optim.Adam.state_dict() layout, parameters in diffusion.parameters() order (DDP:1741, 1793)
    def state_dict(self):
        names = list(self.params)
        state = {}
        if self.opt_step > 0:
            for i, k in enumerate(names):
                state[i] = {'step': torch.tensor(float(self.opt_step)), 'exp_avg': self.m[k], 'exp_avg_sq': self.v[k]}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(len(names)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        names = list(self.params)
        for i, st in sd.get('state', {}).items():
            k = names[int(i)]
            self.m[k].copy_(st['exp_avg'])
            self.v[k].copy_(st['exp_avg_sq'])
            self.opt_step = int(float(st['step']))
        for grp in sd.get('param_groups', [])[:1]:
            self.lr, self.betas, self.eps = grp['lr'], tuple(grp['betas']), grp['eps']

    def _allreduce_mean(self, grads):
        from .distributed import average_gradients
        return average_gradients(grads, lambda flat, sc: ops.affine(flat, sc, 0.))

    def apply(self, grads):
        """clip by global norm, Adam, bump the weight epoch (the sampling engine and UnetTrain re-pack on it)."""
        names = [k for k in self.params if k in grads]
        gl = [grads[k].contiguous() for k in names]
        self.opt_step += 1
        clip = ops.clip_adam_multi_([self.params[k].data for k in names], gl, [self.m[k] for k in names],
                                    [self.v[k] for k in names], self.max_grad_norm, self.lr, self.betas[0], self.betas[1],
                                    self.eps, self.opt_step)
        self.unet._dmh_epoch = getattr(self.unet, '_dmh_epoch', 0) + 1
        self.refresh()
        self.last_norm = clip
        return clip

    def refresh(self):
        """re-pack the training kernels' weight images from the (updated / loaded) parameters"""
        ptrs = tuple(p.data_ptr() for p in self.params.values())
        if self._repack_graph not in (None, False) and ptrs != getattr(self, '_graph_ptrs', None):
            self._repack_graph = None                      # parameter storage moved since the capture: its pointers are stale
        if self._repack_graph is None:
            self.ut.refresh()                              # (re)build: buffers, the pack table, a first fill
            self._graph_ptrs = ptrs
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # thread_local: other threads of the process (e.g. the RCCL watchdog) may keep calling into HIP meanwhile
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self.ut.repack()
                self._repack_graph = g
            except Exception as e:                         # capture not available: stay on plain launches
                print(f'dmhomo_amd: HIP graph capture of the weight re-pack failed ({e}); using plain launches')
                self._repack_graph = False
            return
        if self._repack_graph is False:
            if tuple(p.data_ptr() for p in self.params.values()) != getattr(self, '_built_ptrs', None):
                self.ut.refresh()
            else:
                self.ut.repack()
            self._built_ptrs = tuple(p.data_ptr() for p in self.params.values())
        else:
            self._repack_graph.replay()

    def step(self, batches, draws=None):
        """``batches``: ``accum`` pairs (12-channel batch, classes).  -> summed (loss / accum) like DDP:1849."""
        assert len(batches) == self.accum
        total, acc = None, None
        for i, (img, classes) in enumerate(batches):
            kw = draws[i] if draws is not None else {}
            loss, gr = self.loss_and_grads(img, classes, grad_scale=1.0 / self.accum, **kw)
            acc = gr if acc is None else {k: ops.add(acc[k], gr[k]) for k in acc}
            part = loss / self.accum
            total = part if total is None else total + part
        acc = self._allreduce_mean(acc)
        self.apply(acc)
        return total


class _PLoss(torch.autograd.Function):
    """autograd boundary of GaussianDiffusion.forward: the loss and every parameter gradient are produced together by
    TrainStep.loss_and_grads (HIP kernels, manual tape); backward() just hands the gradients to autograd."""

    @staticmethod
    def forward(ctx, ts, img, classes, *params):
        loss, grads = ts.loss_and_grads(img, classes)
        ctx.grads = [grads.get(k) for k in ts.params]
        return loss.clone()

    @staticmethod
    def backward(ctx, gout):
        s = float(gout)
        out = [None if g is None else (g if s == 1.0 else ops.affine(g, s, 0.)) for g in ctx.grads]
        ctx.grads = None
        return (None, None, None, *out)


def loss_with_grad_fn(diffusion, img, classes):
    ts = diffusion.__dict__.get('_dmh_train_step')
    if ts is None:
        ts = TrainStep(diffusion)
        diffusion.__dict__['_dmh_train_step'] = ts        # not a submodule / parameter: plain attribute
    return _PLoss.apply(ts, img, classes, *ts.params.values())

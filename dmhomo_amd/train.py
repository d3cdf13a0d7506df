"""Training pieces built so far (SURVEY.md §8f "next" row 1 — IN PROGRESS, not a training loop yet).

``ResnetBlockTrain`` runs the forward of one ResnetBlock (CFG:216-241) on the same kernels as sampling while saving what
the backward pass needs, and the backward pass itself on the HIP kernels of csrc/conv_backward.hip and
csrc/norm_backward.hip:

  forward   x -> conv3x3(WS(w1)) -> GN(g1,b1)*(scale+1)+shift -> SiLU -> conv3x3(WS(w2)) -> GN(g2,b2) -> SiLU -> + res(x)
  saved     y1, y2 (conv outputs), the GroupNorm affines coef1/coef2 and (mean, rstd) mr1/mr2  (h = SiLU(GN(y1)) is
            never stored: consumers re-apply the affine + SiLU while staging, exactly as the forward conv does)
  backward  GN+SiLU backward (3 kernels) -> conv weight / bias gradient (fp32 MFMA, pixels as the K axis) -> weight
            standardisation backward -> data gradient = the forward conv kernel with the flipped, transposed weight

Still missing for a training step: the same for attention / LayerNorm / embeddings / up- and down-sampling convs, the loss
gradient (incl. grid_sample wrt its input), Adam + EMA, gradient all-reduce.  ``Trainer.train`` keeps raising.
"""
import torch

from . import ops


class ResnetBlockTrain:
    """one ResnetBlock with raw (trainable) parameters as device tensors.  ``p`` keys: w1 b1 g1 be1 w2 b2 g2 be2 and,
    when the block changes width, rw rb (the 1x1 res_conv).  c1 > 0: the input is cat(x0, x1) (up path)."""

    def __init__(self, p, c0, c1=0, groups=8):
        self.p, self.c0, self.c1, self.groups = p, c0, c1, groups
        self.refresh()

    def refresh(self):
        """(re)pack after a parameter update: standardised weights, forward and data-gradient images."""
        p = self.p
        self.w1s, self.w2s = ops.ws_standardize(p['w1']), ops.ws_standardize(p['w2'])
        cout = p['w1'].shape[0]
        self.cout = cout
        self.f1 = ops.PackedConv(self.w1s, p['b1'], self.c0, self.c1)
        self.f2 = ops.PackedConv(self.w2s, p['b2'], cout)
        self.d1 = ops.conv_dgrad_pack(self.w1s, self.c0 + self.c1)
        self.d2 = ops.conv_dgrad_pack(self.w2s, cout)
        self.fr = self.dr = None
        if 'rw' in p:
            self.fr = ops.PackedConv(p['rw'], p['rb'], self.c0, self.c1)
            self.dr = ops.conv_dgrad_pack(p['rw'], self.c0 + self.c1)

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

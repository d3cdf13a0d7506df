#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Run in the build container only (needs /root/reference, read-only):
    python tests/golden/make_golden.py
The reference is imported with empty stub modules for the packages its dataset /
visualisation / Trainer code wants (cv2, imageio, torchvision, ema_pytorch, the
pip ``denoising_diffusion_pytorch``) — none of them is on the arithmetic path
(SURVEY.md §8c, Appendix B).  Outputs are data only: inputs, recorded RNG draws
and the reference's outputs, as .npz.  Weights are NOT stored: they are rebuilt
from key names by tests/detweights.py, whose checksum is stored instead.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from detweights import det_state_dict, checksum, shapes_of  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    _stub('cv2')
    _stub('imageio')
    tv = _stub('torchvision')
    tv.transforms = _stub('torchvision.transforms')
    tv.utils = _stub('torchvision.utils')
    _stub('ema_pytorch', EMA=object)
    _stub('denoising_diffusion_pytorch')
    _stub('denoising_diffusion_pytorch.version', __version__='0.1.0')
    sys.dont_write_bytecode = True
    sys.path.insert(0, '/root/reference/DGM')
    from denoising_diffusion_models import classifier_free_guidance as cfg
    from denoising_diffusion_models import denoising_diffusion_pytorch as ddp
    return cfg, ddp


def load_det(module, seed=0):
    sd = det_state_dict(shapes_of(module), seed)
    missing = module.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys
    return sd


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.0f} KB')


class RngTape:
    """records every tensor produced by torch.randn / randn_like / Tensor.uniform_."""

    def __enter__(self):
        self.draws = []
        self._randn, self._randn_like, self._uniform = torch.randn, torch.randn_like, torch.Tensor.uniform_
        tape = self

        def randn(*a, **k):
            r = tape._randn(*a, **k)
            tape.draws.append(r.clone())
            return r

        def randn_like(*a, **k):
            r = tape._randn_like(*a, **k)
            tape.draws.append(r.clone())
            return r

        def uniform_(self_, *a, **k):
            r = tape._uniform(self_, *a, **k)
            tape.draws.append(r.clone())
            return r

        torch.randn, torch.randn_like, torch.Tensor.uniform_ = randn, randn_like, uniform_
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like, torch.Tensor.uniform_ = self._randn, self._randn_like, self._uniform


def cond_inputs(B, C, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    rgb_flow = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.4).float()
    return x, rgb_flow, mask


TAPS = ['init_conv', 'downs.0.0', 'downs.0.2', 'downs.1.3', 'downs.3.3', 'mid_block1', 'mid_attn',
        'ups.0.3', 'ups.3.2', 'final_res_block']


def with_taps(model, fn):
    taps, hooks = {}, []
    mods = dict(model.named_modules())
    for name in TAPS:
        hooks.append(mods[name].register_forward_hook(
            lambda m, i, o, name=name: taps.__setitem__(name, o.detach().clone())))
    out = fn()
    for h in hooks:
        h.remove()
    return out, taps


def main():
    torch.set_num_threads(8)
    cfg, ddp = import_reference()
    meta = {'torch': torch.__version__, 'numpy': np.__version__}

    # ---------------------------------------------------------------- F1 unet_cfg_tiny
    torch.manual_seed(0)
    m = cfg.Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1).eval()
    sd = load_det(m)
    meta['unet_cfg_tiny_keys'] = {k: list(s) for k, s in shapes_of(m).items()}
    meta['unet_cfg_tiny_checksum'] = checksum(sd)
    x, rgb_flow, mask = cond_inputs(2, 6, 32, 32, 11)
    t = torch.tensor([999, 37])
    classes = torch.zeros(2, dtype=torch.long)
    with torch.no_grad():
        (out_c, taps) = with_taps(m, lambda: m(x, t, classes, rgb_flow, mask, cond_drop_prob=0.))
        out_n = m(x, t, classes, rgb_flow, mask, cond_drop_prob=1.)
        torch.manual_seed(5)
        keep = torch.zeros(2).float().uniform_(0, 1) < 0.5
        torch.manual_seed(5)
        out_h = m(x, t, classes, rgb_flow, mask)                 # default cond_drop_prob 0.5
        torch.manual_seed(6)
        keep_s = torch.zeros(2).float().uniform_(0, 1) < 0.5
        torch.manual_seed(6)
        out_s = m.forward_with_cond_scale(x, t, classes, rgb_flow=rgb_flow, mask=mask, cond_scale=3.)
    npz('unet_cfg_tiny', x=x, t=t, classes=classes, rgb_flow=rgb_flow, mask=mask, out_cond=out_c,
        out_null=out_n, keep_half=keep, out_half=out_h, keep_scale3=keep_s, out_scale3=out_s,
        **{'tap.' + k: v for k, v in taps.items()})

    # the full-size key list (dim=64) for state_dict compatibility checks
    m64 = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    meta['unet_cfg_dim64_keys'] = {k: list(s) for k, s in shapes_of(m64).items()}
    d64 = cfg.GaussianDiffusion(m64, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0')
    meta['diffusion_cfg_dim64_keys'] = {k: list(s) for k, s in shapes_of(d64).items()}
    del m64, d64

    # ---------------------------------------------------------------- F2 unet_ddp_tiny
    arrays = {}
    for sc in (False, True):
        md = ddp.Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=3, self_condition=sc).eval()
        sdd = load_det(md, seed=1)
        tag = 'sc' if sc else 'nosc'
        meta[f'unet_ddp_tiny_{tag}_keys'] = {k: list(s) for k, s in shapes_of(md).items()}
        meta[f'unet_ddp_tiny_{tag}_checksum'] = checksum(sdd)
        g = torch.Generator().manual_seed(21)
        xd = torch.randn(2, 3, 32, 32, generator=g)
        xs = torch.randn(2, 3, 32, 32, generator=g)
        td = torch.tensor([3, 640])
        with torch.no_grad():
            arrays[f'{tag}.out'] = md(xd, td, xs if sc else None)
            if sc:
                arrays[f'{tag}.out_default'] = md(xd, td)
        arrays.update({'x': xd, 'x_self_cond': xs, 't': td})
    npz('unet_ddp_tiny', **arrays)
    md64 = ddp.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3)
    meta['unet_ddp_dim64_keys'] = {k: list(s) for k, s in shapes_of(md64).items()}
    del md64

    # ---------------------------------------------------------------- F3 blocks_fullwidth
    arrays = {}
    g = torch.Generator().manual_seed(31)
    with torch.no_grad():
        blk = cfg.Block(64, 64).eval()
        load_det(blk, seed=2)
        xb = torch.randn(2, 64, 16, 16, generator=g)
        sc_, sh_ = torch.randn(2, 64, 1, 1, generator=g) * 0.3, torch.randn(2, 64, 1, 1, generator=g) * 0.3
        arrays.update({'block.x': xb, 'block.scale': sc_, 'block.shift': sh_,
                       'block.out': blk(xb, (sc_, sh_)), 'block.out_noss': blk(xb)})
        rb = cfg.ResnetBlock(128, 64, time_emb_dim=256, classes_emb_dim=256).eval()
        load_det(rb, seed=3)
        xr = torch.randn(2, 128, 16, 16, generator=g)
        te, ce = torch.randn(2, 256, generator=g), torch.randn(2, 256, generator=g)
        arrays.update({'resnet.x': xr, 'resnet.t': te, 'resnet.c': ce, 'resnet.out': rb(xr, te, ce)})
        rbi = cfg.ResnetBlock(64, 64, time_emb_dim=256, classes_emb_dim=256).eval()
        load_det(rbi, seed=4)
        arrays.update({'resnet_id.out': rbi(xb, te, ce)})
        la = cfg.Residual(cfg.PreNorm(64, cfg.LinearAttention(64))).eval()
        load_det(la, seed=5)
        arrays.update({'linattn.out': la(xb)})
        at = cfg.Residual(cfg.PreNorm(128, cfg.Attention(128))).eval()
        load_det(at, seed=6)
        xa = torch.randn(2, 128, 8, 8, generator=g)
        arrays.update({'attn.x': xa, 'attn.out': at(xa)})
        dn = cfg.Downsample(64, 128).eval()
        load_det(dn, seed=7)
        up = cfg.Upsample(128, 64).eval()
        load_det(up, seed=8)
        arrays.update({'down.out': dn(xb), 'up.out': up(xa)})
        dnd = ddp.Downsample(64, 128).eval()
        load_det(dnd, seed=9)
        arrays.update({'down_ddp.out': dnd(xb)})
    npz('blocks_fullwidth', **arrays)

    # ---------------------------------------------------------------- F4 schedule
    arrays = {}
    tiny = cfg.Unet(dim=8, channels=6, num_classes=1)
    for sched, T in (('cosine', 1000), ('linear', 1000), ('cosine', 10)):
        d = cfg.GaussianDiffusion(tiny, image_size=16, timesteps=T, beta_schedule=sched)
        for k, v in d.state_dict().items():
            if not k.startswith('model.'):
                arrays[f'{sched}{T}.{k}'] = v
    for S in (4, 32, 250):
        times = torch.linspace(-1, 999, steps=S + 1)
        arrays[f'times{S}'] = np.array(list(reversed(times.int().tolist())), dtype=np.int64)
    npz('schedule', **arrays)

    # ---------------------------------------------------------------- F5 ddim_trace (CFG)
    arrays = {}
    m = cfg.Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1).eval()
    load_det(m)
    _, rgb01, mask = cond_inputs(2, 6, 16, 16, 41)
    rgb01 = (rgb01 + 1) / 2
    flow = torch.randn(2, 2, 16, 16, generator=torch.Generator().manual_seed(42))
    classes = torch.zeros(2, dtype=torch.long)
    for obj in ('pred_x0', 'pred_noise', 'pred_v'):
        d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective=obj)
        steps = []
        orig = d.model_predictions

        def mp(*a, _orig=orig, _steps=steps, **k):
            r = _orig(*a, **k)
            _steps.append(r.pred_x_start.clone())
            return r
        d.model_predictions = mp
        torch.manual_seed(99)
        with RngTape() as tape:
            img, mo, fo = d.sample(classes, rgb01, flow, mask)
        assert len(tape.draws) == 8, len(tape.draws)
        for i, dr in enumerate(tape.draws):
            arrays[f'{obj}.draw{i}'] = dr
        for i, xs_ in enumerate(steps):
            arrays[f'{obj}.x_start{i}'] = xs_
        arrays[f'{obj}.img'] = img
        assert torch.equal(mo, mask) and torch.equal(fo, flow)
    arrays.update({'rgb_flow01': rgb01, 'mask': mask, 'flow': flow, 'classes': classes})
    npz('ddim_trace', **arrays)

    # ---------------------------------------------------------------- F8 train_step forward value (CFG p_losses)
    arrays = {}
    g8 = torch.Generator().manual_seed(61)
    img12 = torch.rand(2, 12, 16, 16, generator=g8)
    img12[:, 6:7] = (img12[:, 6:7] > 0.4).float()
    img12[:, -2:] = (img12[:, -2:] - 0.5) * 6
    tt = torch.tensor([17, 803])
    nz = torch.randn(2, 6, 16, 16, generator=g8)
    arrays.update({'img12': img12, 't': tt, 'noise': nz, 'classes': classes})
    for obj, lt in (('pred_x0', 'l1'), ('pred_noise', 'l2'), ('pred_v', 'l1')):
        d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective=obj, loss_type=lt)
        data, mk, rf, fl = img12[:, :6] * 2 - 1, img12[:, 6:7], img12[:, -5:-2] * 2 - 1, img12[:, -2:]
        torch.manual_seed(7)
        keep = torch.zeros(2).float().uniform_(0, 1) < 0.5
        torch.manual_seed(7)
        with torch.no_grad():
            loss = d.p_losses(data, tt, classes=classes, rgb_flow=rf, flow=fl, mask=mk, noise=nz)
        arrays[f'{obj}.{lt}.loss'] = loss
        arrays[f'{obj}.{lt}.keep'] = keep
    npz('train_forward', **arrays)

    # ---------------------------------------------------------------- F6 ddpm_trace (DDP)
    arrays = {}
    for sc in (False, True):
        tag = 'sc' if sc else 'nosc'
        md = ddp.Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=3, self_condition=sc).eval()
        load_det(md, seed=1)
        d = ddp.GaussianDiffusion(md, image_size=16, timesteps=10, objective='pred_noise')
        torch.manual_seed(7)
        with RngTape() as tape:
            img = d.sample(batch_size=2)
        assert len(tape.draws) == 10
        for i, dr in enumerate(tape.draws):
            arrays[f'{tag}.ddpm.draw{i}'] = dr
        arrays[f'{tag}.ddpm.img'] = img
        d2 = ddp.GaussianDiffusion(md, image_size=16, timesteps=10, sampling_timesteps=4, objective='pred_x0')
        torch.manual_seed(8)
        with RngTape() as tape:
            img = d2.sample(batch_size=2)
        assert len(tape.draws) == 4
        for i, dr in enumerate(tape.draws):
            arrays[f'{tag}.ddim.draw{i}'] = dr
        arrays[f'{tag}.ddim.img'] = img
        if not sc:
            xt = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(9))
            torch.manual_seed(10)
            with RngTape() as tape:
                pi, xs_ = d.p_sample(xt, 5)
            arrays.update({'p_sample.x': xt, 'p_sample.noise': tape.draws[0], 'p_sample.img': pi,
                           'p_sample.x_start': xs_})
    npz('ddpm_trace', **arrays)

    # ---------------------------------------------------------------- F7 geometry
    arrays = {}
    Hs = np.stack([
        np.eye(3),
        np.array([[1, 0, 5.5], [0, 1, -3.25], [0, 0, 1]]),
        np.array([[1.02, 0.01, -4], [-0.015, 0.98, 6], [1e-5, -2e-5, 1]]),
        np.array([[0.97, -0.03, 8], [0.025, 1.03, -7.5], [-3e-5, 2.5e-5, 1]]),
        np.array([[1.1, 0.08, -20], [-0.06, 0.9, 15], [2e-4, 1e-4, 1]]),
        np.array([[0.8, 0.1, 40], [0.12, 1.2, -35], [-4e-4, 3e-4, 1]]),
    ])
    arrays['H0'] = Hs
    for tag, (h, w) in (('a', (128, 128)), ('b', (32, 48))):
        H1 = np.stack([ddp.adapt_homography_to_preprocessing_v3(360, 640, Hm, h, w) for Hm in Hs])
        flows = np.stack([ddp.homo_to_flow(Hm[None, None], h, w) for Hm in H1])        # (6,h,w,2) f32
        rgbs = np.stack([ddp.flow_to_image(f) for f in flows])
        arrays.update({f'{tag}.H1': H1, f'{tag}.flow': flows, f'{tag}.rgb': rgbs})
        ft = torch.from_numpy(flows).permute(0, 3, 1, 2).contiguous()
        img = torch.rand(6, 3, h, w, generator=torch.Generator().manual_seed(50))
        arrays[f'{tag}.img'] = img
        arrays[f'{tag}.warp'] = ddp.flow_warp(img, ft)
        if tag == 'b':
            homos = ddp.homo_gen(ft)
            arrays[f'{tag}.homo_gen'] = homos
            ret = ddp.saveTrainPair(torch.cat([img, img.flip(0)], 1), None, ft)
            arrays[f'{tag}.pair_imgs'] = ret['imgs']
            arrays[f'{tag}.pair_homos'] = ret['homos']
    # a random (non-homography) flow exercises the border clamp of grid_sample
    fr = torch.randn(2, 2, 24, 40, generator=torch.Generator().manual_seed(51)) * 9
    ir = torch.rand(2, 5, 24, 40, generator=torch.Generator().manual_seed(52))
    arrays.update({'r.flow': fr, 'r.img': ir, 'r.warp': ddp.flow_warp(ir, fr)})
    npz('geometry', **arrays)

    with open(os.path.join(HERE, 'meta.json'), 'w') as f:
        json.dump(meta, f, indent=0, sort_keys=True)
    print('meta.json')


if __name__ == '__main__':
    main()

"""-m gpu: condition builder & geometry kernels (K7-K9) against the golden vectors of the reference.
Bit-exact contract: grid-sample corner indices (int32) and the uint8 export."""
import os

import numpy as np
import pytest
import torch

from gpu_util import dev, close, report
from oracle import geometry as OG

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gd(golden_dir):
    return {k: v for k, v in np.load(os.path.join(golden_dir, 'geometry.npz')).items()}


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize('tag,h,w', [('a', 128, 128), ('b', 32, 48)])
def test_homography_flow_rgb(gd, tag, h, w):
    from dmhomo_amd import ddpm, geometry
    flow, rgb = ddpm.homo_to_flow_rgb(gd[f'{tag}.H1'], h, w)
    want_flow = T(gd[f'{tag}.flow']).permute(0, 3, 1, 2)
    want_rgb = T(gd[f'{tag}.rgb']).permute(0, 3, 1, 2)
    # f64 matvec + divide, cast to fp32: identical up to (rare) 1-ulp f64 differences before the cast
    close(f'flow {tag}', flow.cpu(), want_flow, rtol=0, atol=4e-6)
    exact = (flow.cpu() == want_flow).float().mean().item()
    print(f'[parity] flow {tag}: bit-exact fraction {exact:.6f}')
    assert exact > 0.99
    close(f'rgb {tag}', rgb.cpu(), want_rgb, rtol=0, atol=2e-5)     # atan2f / HSV sextants in fp32
    one = ddpm.homo_to_flow(gd[f'{tag}.H1'][3][None, None], h, w)
    assert one.shape == (h, w, 2) and one.dtype == np.float32
    np.testing.assert_allclose(one, gd[f'{tag}.flow'][3], rtol=0, atol=4e-6)
    img = geometry.flow_to_image(gd[f'{tag}.flow'][4])
    np.testing.assert_allclose(img, gd[f'{tag}.rgb'][4], rtol=0, atol=2e-5)
    a = ddpm.adapt_homography_to_preprocessing_v3(360, 640, gd['H0'][2], h, w)
    np.testing.assert_allclose(a, gd[f'{tag}.H1'][2], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize('tag', ['a', 'b', 'r'])
def test_flow_warp(gd, tag):
    from dmhomo_amd import ops, ddpm
    if tag == 'r':
        img, flow = T(gd['r.img']), T(gd['r.flow'])
    else:
        img, flow = T(gd[f'{tag}.img']), T(gd[f'{tag}.flow']).permute(0, 3, 1, 2).contiguous()
    out, x0, y0 = ops.flow_warp(img.to(dev()), flow.to(dev()), want_indices=True)
    ix, iy, rx0, ry0 = OG.warp_coords(flow)
    assert torch.equal(x0.cpu(), rx0) and torch.equal(y0.cpu(), ry0)          # bit-exact corner indices
    want = T(gd[f'{tag}.warp'])
    report(f'warp {tag}', out.cpu(), want)
    assert torch.equal(out.cpu(), want)                                        # same fma chain as torch's CPU kernel
    assert torch.equal(ddpm.flow_warp(img.to(dev()), flow.to(dev())), out)
    assert torch.equal(out.cpu(), OG.flow_warp(img, flow))


def test_flow_warp_known_answers():
    from dmhomo_amd import ops
    img = torch.rand(1, 2, 5, 9, generator=torch.Generator().manual_seed(1))
    out = ops.flow_warp(img.to(dev()), torch.zeros(1, 2, 5, 9, device=dev()))
    assert torch.equal(out.cpu(), img)                                         # W-1, H-1 powers of two: exact identity
    out, x0, y0 = ops.flow_warp(img.to(dev()), torch.full((1, 2, 5, 9), 100., device=dev()), want_indices=True)
    assert int(x0.min()) == 8 and int(y0.min()) == 4                            # border clamp
    assert torch.equal(out.cpu(), img[:, :, 4:5, 8:9].expand(1, 2, 5, 9))


def test_dlt_homography_and_save_train_pair(gd):
    from dmhomo_amd import ddpm
    ft = T(gd['b.flow']).permute(0, 3, 1, 2).contiguous()
    hg = ddpm.homo_gen(ft.to(dev()))
    assert hg.shape == (6, 1, 3, 3) and hg.dtype == torch.float64
    # same least-squares problem as the reference's pinv, solved through f64 normal equations
    np.testing.assert_allclose(hg.cpu().numpy(), gd['b.homo_gen'], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(hg.cpu().numpy()[:, 0], gd['b.H1'], rtol=2e-4, atol=2e-4)      # H -> flow -> H
    img = T(gd['b.img'])
    pair = torch.cat([img, img.flip(0)], 1)
    ret = ddpm.saveTrainPair(pair.to(dev()), None, ft.to(dev()))
    assert ret['imgs'].dtype == np.uint8 and np.array_equal(ret['imgs'], gd['b.pair_imgs'])   # bit-exact
    np.testing.assert_allclose(ret['homos'], gd['b.pair_homos'], rtol=1e-6, atol=1e-8)
    # full-size round trip (128x128): identity and a strong perspective
    fa = T(gd['a.flow']).permute(0, 3, 1, 2).contiguous()
    ha = ddpm.homo_gen(fa.to(dev())).cpu().numpy()[:, 0]
    np.testing.assert_allclose(ha, gd['a.H1'], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(ha, OG.homo_gen_normal_eq(fa).numpy()[:, 0], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize('seed', range(6))
def test_flow_warp_fuzz_bit_exact(seed):
    """random shapes (odd sizes, 1-6 channels, batch 1-3) and flows from sub-pixel to several image widths, incl. exact
    integers, exact halves and the image border: corner indices and the warped image bit-exact against the oracle (torch's CPU
    grid_sample, align_corners=True, border padding — what the reference's flow_warp calls, DDP:1262-1280)"""
    from dmhomo_amd import ops
    import random
    rnd = random.Random(seed)
    gen = torch.Generator().manual_seed(1000 + seed)
    B, C = rnd.randint(1, 3), rnd.choice([1, 2, 3, 6])
    H, W = rnd.randint(2, 70), rnd.randint(2, 70)
    img = torch.randn(B, C, H, W, generator=gen)
    mag = rnd.choice([0.4, 3.0, 40.0, 300.0])
    flow = (torch.rand(B, 2, H, W, generator=gen) * 2 - 1) * mag
    sel = torch.rand(B, 2, H, W, generator=gen)
    flow = torch.where(sel < 0.15, flow.round(), flow)                          # integer displacements
    flow = torch.where((sel >= 0.15) & (sel < 0.25), flow.round() + 0.5, flow)  # exact halves
    flow[:, :, 0, :] = -1.0                                                      # rows that leave through the top border
    out, x0, y0 = ops.flow_warp(img.to(dev()), flow.to(dev()), want_indices=True)
    ix, iy, rx0, ry0 = OG.warp_coords(flow)
    assert torch.equal(x0.cpu(), rx0) and torch.equal(y0.cpu(), ry0)
    assert torch.equal(out.cpu(), OG.flow_warp(img, flow))


def test_dlt_recovers_homographies_of_any_strength():
    """flow of a known homography -> dmh_dlt_homography -> the homography (f64, normal equations): identity (zero flow), pure
    translation, strong perspective; and the batched records of saveTrainPair"""
    from dmhomo_amd import ddpm
    Hs = np.stack([np.eye(3),
                   np.array([[1, 0, 7.5], [0, 1, -3.25], [0, 0, 1.]]),
                   np.array([[1.05, .02, 2.], [-.03, .97, 1.], [4e-4, -3e-4, 1.]]),
                   np.array([[.9, .1, -4.], [.05, 1.1, 6.], [-6e-4, 5e-4, 1.]])])
    for S in (32, 128):
        flow, _ = ddpm.homo_to_flow_rgb(Hs, S, S)
        got = ddpm.homo_gen(flow).reshape(-1, 3, 3).cpu().numpy()
        err = np.abs(got - Hs).max()
        print(f'[parity] DLT round trip at {S}x{S}: max |H - H_true| = {err:.2e}')
        assert err < 8e-5        # measured 7.5e-6: the flow is fp32 (1e-7 relative on coordinates up to 128) and w' carries the reference's + 1e-6


@pytest.mark.parametrize('pad', ['border', 'zeros', 'reflection'])
@pytest.mark.parametrize('mode', ['bilinear', 'nearest', 'bicubic'])
def test_flow_warp_padding_and_mode_variants(golden_dir, pad, mode):
    """flow_warp(x, flow, pad, mode) DDP:1262-1280 beyond its defaults, as it forwards them to F.grid_sample: the reference's
    outputs (tests/golden/make_golden_r5.py: targets outside the image, half-pixel ties) — bit for bit for every combination
    but 'reflection' (fmodf of the reflected coordinate: <= 1e-6) — and a fuzz against the oracle's explicit taps"""
    from dmhomo_amd import ddpm
    r5 = {k: v for k, v in np.load(os.path.join(golden_dir, 'r5.npz')).items()}
    got = ddpm.flow_warp(T(r5['warp.x']).to(dev()), T(r5['warp.flow']).to(dev()), pad=pad, mode=mode).cpu()
    want = T(r5[f'warp.{pad}.{mode}'])
    report(f'warp {pad} {mode}', got, want)
    if mode == 'bicubic':
        assert float((got - want).abs().max()) <= 1e-5      # 16 taps, cubic weights up to 1: sums in another order
    elif pad == 'reflection':
        assert float((got - want).abs().max()) <= 1e-6
    else:
        assert torch.equal(got, want)
    gen = torch.Generator().manual_seed(77)
    for H, W, mag in ((7, 9, 3.0), (33, 17, 40.0), (2, 2, 5.0)):
        img = torch.randn(2, 3, H, W, generator=gen)
        flow = (torch.rand(2, 2, H, W, generator=gen) * 2 - 1) * mag
        flow[:, :, 0, :] = flow[:, :, 0, :].round() + 0.5
        got = ddpm.flow_warp(img.to(dev()), flow.to(dev()), pad=pad, mode=mode).cpu()
        ref = OG.flow_warp_general(img, flow, pad, mode)
        err = float((got - ref).abs().max())
        assert err <= (2e-5 if (pad == 'reflection' or mode == 'bicubic') else 0.0), (pad, mode, H, W, err)
    with pytest.raises(NotImplementedError):
        ddpm.flow_warp(img.to(dev()), flow.to(dev()), mode='trilinear')

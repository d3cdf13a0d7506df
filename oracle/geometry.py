"""Oracle: condition builder & geometry (test infrastructure, CPU numpy/torch).

Restates the helpers of DGM/denoising_diffusion_models/denoising_diffusion_pytorch.py
(tag DDP) that build the flow condition and consume the sampled pairs:
  G1 adapt_homography_to_preprocessing_v3  DDP:978-988
  G2 mesh_grid_np / get_flow_np / homo_to_flow  DDP:913-975
  G3 flow_to_image (+ matplotlib.colors.hsv_to_rgb, matplotlib 3.10.8)  DDP:1471-1486
  G4 flow_warp / mesh_grid / norm_grid (+ torch grid_sample, bilinear, border,
     align_corners=True; ATen/native/cpu/GridSamplerKernel.cpp)  DDP:1262-1299
  G5 get_grid / DLT_solve / homo_gen  DDP:1558-1661
  G6 saveTrainPair  DDP:1664-1678
"""
import numpy as np
import torch


# ------------------------------------------------------------------ G1
def adapt_homography(h0, w0, H, h1, w1):
    """H1 = M1 (M0^-1 H M0) M1^-1, float64, DDP:978-988."""
    def m(h, w):
        return np.array([[w / 2.0, 0., w / 2.0], [0., h / 2.0, h / 2.0], [0., 0., 1.]])
    M0, M1 = m(h0, w0), m(h1, w1)
    Hn = np.matmul(np.matmul(np.linalg.inv(M0), H), M0)
    return np.matmul(np.matmul(M1, Hn), np.linalg.inv(M1))


# ------------------------------------------------------------------ G2
def homo_to_flow(homo, H, W):
    """single-homography path (``divide`` == 1) of get_flow_np, DDP:927-975.

    p' = Hm [x, y, 1]^T in float64 over an int64 pixel grid, w' += 1e-6
    unconditionally (DDP:958-959), flow = p'/w' - p, cast to float32, (H, W, 2)."""
    Hm = np.asarray(homo, dtype=np.float64).reshape(3, 3)
    xs = np.arange(W)[None, :].repeat(H, 0)
    ys = np.arange(H)[:, None].repeat(W, 1)
    p = np.stack([xs, ys, np.ones_like(xs)], -1)[..., None]          # (H,W,3,1) int64
    q = np.matmul(np.broadcast_to(Hm, (H, W, 3, 3)), p)[..., 0]      # (H,W,3) f64
    wv = q[..., 2] + 1e-6
    fx = q[..., 0] / wv - xs
    fy = q[..., 1] / wv - ys
    return np.stack([fx, fy], -1).astype(np.float32)


# ------------------------------------------------------------------ G3
def hsv_to_rgb(hsv):
    """matplotlib.colors.hsv_to_rgb restated (sextant select), dtype preserved."""
    h, s, v = hsv[..., 0], hsv[..., 1], hsv[..., 2]
    i = (h * 6.0).astype(int)
    f = (h * 6.0) - i
    p = v * (1.0 - s)
    q = v * (1.0 - s * f)
    t = v * (1.0 - s * (1.0 - f))
    r = np.empty_like(h)
    g = np.empty_like(h)
    b = np.empty_like(h)
    for k, (rr, gg, bb) in enumerate(((v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q))):
        m = (i % 6 == k)
        r[m], g[m], b[m] = rr[m], gg[m], bb[m]
    m = (s == 0)
    r[m], g[m], b[m] = v[m], v[m], v[m]
    return np.stack([r, g, b], -1)


def flow_to_image(flow, max_flow=256):
    """DDP:1471-1486: hue = angle, saturation = 8*|flow|/max_flow clipped, value = 1."""
    max_flow = max(max_flow, 1.) if max_flow is not None else np.max(flow)
    n = 8
    u, v = flow[:, :, 0], flow[:, :, 1]
    mag = np.sqrt(np.square(u) + np.square(v))
    ang = np.arctan2(v, u)
    im_h = np.mod(ang / (2 * np.pi) + 1, 1)
    im_s = np.clip(mag * n / max_flow, a_min=0, a_max=1)
    im_v = np.clip(n - im_s, a_min=0, a_max=1)
    return hsv_to_rgb(np.stack([im_h, im_s, im_v], 2))


# ------------------------------------------------------------------ G4
def warp_coords(flow12):
    """fp32 source coordinates of flow_warp, with the exact op order of the
    reference + torch's CPU grid sampler:

      v  = base + flow                                   DDP:1267
      g  = 2.0 * v / (W-1) - 1.0                         DDP:1297-1298
      ix = (g + 1) * ((W-1)/2)      (align_corners)      GridSamplerKernel.cpp
      ix = min(W-1, max(ix, 0))     (border)
      x0 = floor(ix)

    flow12: (B,2,H,W) fp32.  Returns ix, iy (fp32) and x0, y0 (int32)."""
    B, _, H, W = flow12.shape
    xs = torch.arange(0, W).repeat(B, H, 1).to(flow12.dtype)
    ys = torch.arange(0, H).repeat(B, W, 1).transpose(1, 2).to(flow12.dtype)
    vx = xs + flow12[:, 0]
    vy = ys + flow12[:, 1]
    gx = 2.0 * vx / (W - 1) - 1.0
    gy = 2.0 * vy / (H - 1) - 1.0
    sx = torch.tensor((W - 1) / 2, dtype=flow12.dtype)
    sy = torch.tensor((H - 1) / 2, dtype=flow12.dtype)
    ix = ((gx + 1) * sx).clamp(min=0).clamp(max=W - 1)
    iy = ((gy + 1) * sy).clamp(min=0).clamp(max=H - 1)
    return ix, iy, ix.floor().to(torch.int32), iy.floor().to(torch.int32)


def flow_warp(x, flow12):
    """flow_warp(x, flow12, pad='border', mode='bilinear'), DDP:1262-1280.

    4-tap bilinear with weights w = ix - x0, e = (x0+1) - ix, n = iy - y0,
    s = (y0+1) - iy and the accumulation of torch's vectorised CPU kernel, which
    the golden vectors pin bit-for-bit as an FMA chain
        acc = nw*(s*e); acc = fma(ne, s*w, acc); acc = fma(sw, n*e, acc); acc = fma(se, n*w, acc)
    (emulated here through float64: an fp32 product is exact in f64).  Corners
    beyond the image carry zero weight under border padding."""
    B, C, H, W = x.shape
    ix, iy, x0, y0 = warp_coords(flow12)
    w = ix - x0
    e = (x0 + 1) - ix
    n = iy - y0
    s = (y0 + 1) - iy
    x0l, y0l = x0.long(), y0.long()
    x1l = (x0l + 1).clamp(max=W - 1)
    y1l = (y0l + 1).clamp(max=H - 1)
    in_x1 = ((x0l + 1) <= W - 1)
    in_y1 = ((y0l + 1) <= H - 1)
    flat = x.reshape(B, C, H * W)

    def gather(yy, xx):
        idx = (yy * W + xx).reshape(B, 1, H * W).expand(B, C, H * W)
        return flat.gather(2, idx).reshape(B, C, H, W)

    zero = torch.zeros((), dtype=x.dtype)
    nw = gather(y0l, x0l)
    ne = torch.where(in_x1[:, None], gather(y0l, x1l), zero)
    sw = torch.where(in_y1[:, None], gather(y1l, x0l), zero)
    se = torch.where((in_x1 & in_y1)[:, None], gather(y1l, x1l), zero)
    acc = nw * (s * e)[:, None]
    for val, wt in ((ne, s * w), (sw, n * e), (se, n * w)):
        acc = (val.double() * wt[:, None].double() + acc.double()).to(x.dtype)
    return acc


def flow_warp_general(x, flow12, pad='border', mode='bilinear'):
    """flow_warp(x, flow12, pad, mode), DDP:1262-1280, for the other values it forwards to F.grid_sample
    (align_corners=True): pad 'border' | 'zeros' | 'reflection', mode 'bilinear' | 'nearest' | 'bicubic'.  Coordinates as warp_coords
    up to the padding rule (ATen/native/GridSampler.h: reflect_coordinates(in, 0, 2 (size - 1)) then clip_coordinates for
    'reflection'; no clip for 'zeros', whose taps outside the image read 0); 'nearest' rounds ties to even (nearbyint)."""
    B, C, H, W = x.shape
    dt = flow12.dtype
    xs = torch.arange(0, W).repeat(B, H, 1).to(dt)
    ys = torch.arange(0, H).repeat(B, W, 1).transpose(1, 2).to(dt)
    gx = 2.0 * (xs + flow12[:, 0]) / (W - 1) - 1.0
    gy = 2.0 * (ys + flow12[:, 1]) / (H - 1) - 1.0

    def coord(g, size):
        i = (g + 1) * torch.tensor((size - 1) / 2, dtype=dt)
        if pad == 'reflection':
            span = torch.tensor(float(size - 1), dtype=dt)
            a = i.abs()
            extra = torch.fmod(a, span)
            flips = torch.floor(a / span)
            i = torch.where(torch.fmod(flips, 2) == 0, extra, span - extra)
        if pad != 'zeros':
            i = i.clamp(min=0).clamp(max=size - 1)
        return i
    flat = x.reshape(B, C, H * W)
    zero = torch.zeros((), dtype=x.dtype)
    if mode == 'bicubic':
        # cubic convolution (A = -0.75) over the 4 x 4 taps around floor of the UNPADDED coordinate; the padding rule applies
        # to each tap index (get_value_bounded); rows are interpolated along x first, then along y
        def pad_idx(i, size):
            if pad == 'reflection':
                span = torch.tensor(float(size - 1), dtype=dt)
                a = i.abs()
                extra = torch.fmod(a, span)
                i = torch.where(torch.fmod(torch.floor(a / span), 2) == 0, extra, span - extra)
            return i if pad == 'zeros' else i.clamp(min=0).clamp(max=size - 1)

        def coeffs(t):
            A = -0.75
            c1 = lambda v: ((A * v - 5 * A) * v + 8 * A) * v - 4 * A
            c2 = lambda v: ((A + 2) * v - (A + 3)) * v * v + 1
            return [c1(t + 1), c2(t), c2(1 - t), c1(2 - t)]
        ux = (gx + 1) * torch.tensor((W - 1) / 2, dtype=dt)
        uy = (gy + 1) * torch.tensor((H - 1) / 2, dtype=dt)
        x0, y0 = ux.floor(), uy.floor()
        cx, cy = coeffs(ux - x0), coeffs(uy - y0)
        acc = torch.zeros_like(x)
        for i in range(4):
            py = pad_idx(y0 - 1 + i, H)
            row = torch.zeros_like(x)
            for j in range(4):
                px = pad_idx(x0 - 1 + j, W)
                ok = (px >= 0) & (px <= W - 1) & (py >= 0) & (py <= H - 1)
                idx = (py.clamp(0, H - 1).long() * W + px.clamp(0, W - 1).long()).reshape(B, 1, H * W).expand(B, C, H * W)
                v = torch.where(ok[:, None], flat.gather(2, idx).reshape(B, C, H, W), zero)
                row = row + v * cx[j][:, None]
            acc = acc + row * cy[i][:, None]
        return acc
    ix, iy = coord(gx, W), coord(gy, H)

    def tap(fx, fy):
        ok = (fx >= 0) & (fx <= W - 1) & (fy >= 0) & (fy <= H - 1)
        idx = (fy.clamp(0, H - 1).long() * W + fx.clamp(0, W - 1).long()).reshape(B, 1, H * W).expand(B, C, H * W)
        return torch.where(ok[:, None], flat.gather(2, idx).reshape(B, C, H, W), zero)
    if mode == 'nearest':
        return tap(torch.round(ix), torch.round(iy))           # torch.round: half to even, as nearbyint
    x0, y0 = ix.floor(), iy.floor()
    w, e, n, s_ = ix - x0, (x0 + 1) - ix, iy - y0, (y0 + 1) - iy
    acc = tap(x0, y0) * (s_ * e)[:, None]
    for val, wt in ((tap(x0 + 1, y0), s_ * w), (tap(x0, y0 + 1), n * e), (tap(x0 + 1, y0 + 1), n * w)):
        acc = (val.double() * wt[:, None].double() + acc.double()).to(x.dtype)   # one fma per tap, as flow_warp above
    return acc


# ------------------------------------------------------------------ G5
def dlt_system(flow):
    """rows of the DLT system of DLT_solve, DDP:1612-1637, for one H per sample:
    every pixel (x, y) -> (x', y') = (x, y) + flow gives
       [x y 1 0 0 0 -x'x -x'y] h = x'
       [0 0 0 x y 1 -y'x -y'y] h = y'
    flow: (B,2,H,W).  Returns A (B, 2HW, 8), b (B, 2HW, 1) in float64."""
    B, _, H, W = flow.shape
    xs = torch.arange(W, dtype=torch.float64).view(1, 1, W).expand(B, H, W).reshape(B, -1)
    ys = torch.arange(H, dtype=torch.float64).view(1, H, 1).expand(B, H, W).reshape(B, -1)
    # grid is float32 -> float64; the sum with the fp32 flow is done in float64 (DDP:1655,1619)
    xd = xs + flow[:, 0].reshape(B, -1).to(torch.float64)
    yd = ys + flow[:, 1].reshape(B, -1).to(torch.float64)
    one = torch.ones_like(xs)
    zero = torch.zeros_like(xs)
    ru = torch.stack([xs, ys, one, zero, zero, zero, -xd * xs, -xd * ys], -1)
    rv = torch.stack([zero, zero, zero, xs, ys, one, -yd * xs, -yd * ys], -1)
    A = torch.stack([ru, rv], 2).reshape(B, -1, 8)
    b = torch.stack([xd, yd], 2).reshape(B, -1, 1)
    return A, b


def homo_gen(flow):
    """homo_gen / DLT_solve, DDP:1647-1661,1639-1643: h = pinv(A) b, H = [h, 1] (B,1,3,3) f64."""
    A, b = dlt_system(flow)
    h8 = torch.matmul(torch.linalg.pinv(A), b).reshape(-1, 8)
    Hm = torch.cat((h8, torch.ones(h8.shape[0], 1, dtype=h8.dtype)), 1)
    return Hm.reshape(-1, 1, 3, 3)


def homo_gen_normal_eq(flow):
    """same least-squares problem through column-equilibrated normal equations
    (what the HIP K9 kernel computes): h = D (D A^T A D)^-1 D A^T b."""
    A, b = dlt_system(flow)
    G = A.transpose(1, 2) @ A
    r = A.transpose(1, 2) @ b
    d = 1.0 / torch.sqrt(torch.diagonal(G, dim1=1, dim2=2))
    Gs = G * d[:, :, None] * d[:, None, :]
    hs = torch.linalg.solve(Gs, r * d[:, :, None])
    h8 = (hs * d[:, :, None]).reshape(-1, 8)
    Hm = torch.cat((h8, torch.ones(h8.shape[0], 1, dtype=h8.dtype)), 1)
    return Hm.reshape(-1, 1, 3, 3)


# ------------------------------------------------------------------ G6
def save_train_pair(imgs, flows):
    """saveTrainPair, DDP:1664-1678: uint8 by truncation of img*255 (fp32), homographies f64."""
    assert torch.max(imgs) <= 1
    imgs_np = (imgs.detach().cpu().numpy() * 255).astype(np.uint8)
    homos = homo_gen(flows).detach().cpu().numpy().squeeze()
    return {'imgs': imgs_np, 'homos': homos}

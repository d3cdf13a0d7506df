"""ORACLE (test infrastructure only) for the condition dataset path, SURVEY.md §8f row 4.

CPU restatement in numpy of what UnHomoTrainData.__getitem__ (DDP:1097-1163) computes per item with OpenCV:

    img  = cv2.resize(cv2.imread(png).astype(np.float32) / 255., (S, S))                         DDP:1118-1123
    mask = cv2.dilate(cv2.erode(cv2.resize(mask, (S, S), interpolation=cv2.INTER_NEAREST),
                                np.ones((3, 3), np.uint8)), np.ones((3, 3), np.uint8))           DDP:1129-1133
    item = concat(img1, img2, mask, flow_to_image(flow), flow)  -> CHW float32                   DDP:1157-1163

PARITY UNPINNED for the OpenCV part: cv2 (opencv-python, unpinned in the reference's requirements) is not installed in
the build image, and the reference holds no fixture for this path.  What is restated is OpenCV's published algorithm
(modules/imgproc/src/resize.cpp): INTER_LINEAR maps pixel centres, fx = float((dx + 0.5) * scale - 0.5), taps clamped
to the image with weight 0 on the clamped side, a horizontal pass then a vertical pass in float32; INTER_NEAREST takes
sx = min(floor(dx * (1 / (dst / src))), src - 1); erode / dilate with a 3x3 box ignore pixels outside the image (the
default border value is +/- max).  OpenCV's SIMD build may fuse the multiply-adds; the tests allow 1e-6 for that.
The flow / HSV half of the item is oracle/geometry.py, which IS pinned by the reference's golden vectors.
"""
import numpy as np

from . import geometry as G


def _lin_taps(nd, ns):
    scale = float(ns) / nd
    f = ((np.arange(nd, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= ns - 1
    f[hi], s[hi] = 0.0, ns - 1
    return s, np.minimum(s + 1, ns - 1), (np.float32(1.0) - f).astype(np.float32), f.astype(np.float32)


def resize_linear(img, hd, wd):
    """cv2.resize(img, (wd, hd)) for float32 HxWxC (INTER_LINEAR)."""
    img = np.asarray(img, dtype=np.float32)
    hs, ws = img.shape[:2]
    x0, x1, a0, a1 = _lin_taps(wd, ws)
    y0, y1, b0, b1 = _lin_taps(hd, hs)
    rows = img[:, x0] * a0[None, :, None] + img[:, x1] * a1[None, :, None]          # horizontal pass
    return (rows[y0] * b0[:, None, None] + rows[y1] * b1[:, None, None]).astype(np.float32)


def resize_nearest(m, hd, wd):
    """cv2.resize(m, (wd, hd), interpolation=cv2.INTER_NEAREST) for HxW."""
    hs, ws = m.shape
    sx = np.minimum(np.floor(np.arange(wd) * (1.0 / (float(wd) / ws))).astype(np.int64), ws - 1)
    sy = np.minimum(np.floor(np.arange(hd) * (1.0 / (float(hd) / hs))).astype(np.int64), hs - 1)
    return m[sy][:, sx]


def _box3(m, fn, fill):
    h, w = m.shape
    p = np.full((h + 2, w + 2), fill, dtype=m.dtype)
    p[1:-1, 1:-1] = m
    out = p[1:-1, 1:-1].copy()
    for dy in range(3):
        for dx in range(3):
            out = fn(out, p[dy:dy + h, dx:dx + w])
    return out


def erode3(m):
    return _box3(m, np.minimum, np.inf)


def dilate3(m):
    return _box3(m, np.maximum, -np.inf)


def build_item(img1_u8, img2_u8, mask, homo_f, size):
    """the (12, S, S) float32 item of DDP:1162; img*_u8: HxWx3 uint8 as cv2.imread returns them (BGR)."""
    i1 = resize_linear(img1_u8.astype(np.float32) / 255., size, size)
    i2 = resize_linear(img2_u8.astype(np.float32) / 255., size, size)
    mk = dilate3(erode3(resize_nearest(np.squeeze(mask).astype(np.float32), size, size)))[:, :, None]
    homo = G.adapt_homography(360, 640, np.asarray(homo_f, dtype=np.float64), size, size)
    flow = G.homo_to_flow(homo, size, size)                      # DDP:1157
    rgb = G.flow_to_image(flow).astype(np.float32)               # DDP:1160
    return np.concatenate((i1, i2, mk, rgb, flow), axis=2).transpose(2, 0, 1).astype(np.float32)

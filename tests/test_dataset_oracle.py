"""CPU: the oracle's restatement of the OpenCV preprocessing (oracle/dataset.py; cv2 itself is absent, parity with it is
unpinned) against independent implementations of the same published definitions: torch's bilinear / nearest
interpolation (half-pixel centres, clamped taps; floor(dx*scale)) and scipy's grey morphology."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import dataset as OD


def _img(h, w, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def test_resize_linear_matches_torch_half_pixel_bilinear():
    for (h, w, s) in ((360, 640, 128), (90, 70, 256), (128, 128, 128), (37, 53, 16)):
        img = _img(h, w, h + w).astype(np.float32) / 255.
        got = OD.resize_linear(img, s, s)
        ref = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(s, s), mode='bilinear',
                            align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(got - ref).max() < 2e-6, (h, w, s)


def test_resize_nearest_matches_torch_nearest():
    for (h, w, s) in ((360, 640, 128), (45, 80, 256), (128, 128, 128)):
        m = (np.random.default_rng(h).random((h, w)) > 0.5).astype(np.float32)
        got = OD.resize_nearest(m, s, s)
        ref = F.interpolate(torch.from_numpy(m)[None, None], size=(s, s), mode='nearest')[0, 0].numpy()
        assert np.array_equal(got, ref), (h, w, s)


def test_opening_matches_scipy_grey_morphology():
    from scipy import ndimage
    m = (np.random.default_rng(3).random((64, 48)) > 0.35).astype(np.float32)
    e = ndimage.grey_erosion(m, size=(3, 3), mode='constant', cval=np.inf)
    d = ndimage.grey_dilation(e, size=(3, 3), mode='constant', cval=-np.inf)
    assert np.array_equal(OD.erode3(m), e)
    assert np.array_equal(OD.dilate3(OD.erode3(m)), d)


def test_build_item_layout():
    it = OD.build_item(_img(360, 640, 1), _img(360, 640, 2), np.ones((1, 360, 640)), np.eye(3), 32)
    assert it.shape == (12, 32, 32) and it.dtype == np.float32
    assert np.all(it[6] == 1.0) and np.abs(it[10:]).max() < 1e-4            # identity homography: no flow

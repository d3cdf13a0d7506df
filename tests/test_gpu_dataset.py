"""-m gpu: the condition dataset path — UnHomoTrainData on the device kernels against the oracle's
restatement of the reference's OpenCV preprocessing (oracle/dataset.py), on a small synthetic copy of the CA-Homo layout."""
import os

import numpy as np
import pytest
import torch

from gpu_util import dev
from oracle import dataset as OD

pytestmark = pytest.mark.gpu


def _make_dataset(root, n=5, hw=(360, 640), seed=0, ragged=False):
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, 'HomoGAN_Bug_Masks'))
    labels, truth = {}, {}
    hw0 = hw
    for i in range(n):
        hw = (hw0[0] - 24 * (i % 3), hw0[1] - 40 * (i % 2)) if ragged else hw0      # mixed source sizes in one batch
        d = f'{i:04d}'
        os.makedirs(os.path.join(root, d), exist_ok=True)
        a, b = f'{d}_{10000 + i}', f'{d}_{10001 + i}'
        name = a + '_' + b
        imgs = []
        for nm in (a, b):
            # smooth + noisy content so that the bilinear taps matter
            yy, xx = np.mgrid[0:hw[0], 0:hw[1]]
            base = 127 + 100 * np.sin(xx / (7.0 + i) + rng.random()) * np.cos(yy / (5.0 + i))
            rgb = np.clip(base[:, :, None] + rng.integers(-20, 20, size=hw + (3,)), 0, 255).astype(np.uint8)
            Image.fromarray(rgb).save(os.path.join(root, d, nm + '.png'))
            imgs.append(rgb[:, :, ::-1])                              # what cv2.imread returns: BGR
        mask = (rng.random((1,) + hw) > 0.3).astype(np.float64)
        mask[0, 40:300, 100:500] = 1.0
        np.save(os.path.join(root, 'HomoGAN_Bug_Masks', name + '.npy'), mask)
        Hf = np.eye(3) + np.array([[.02, -.01, 6.], [.015, -.02, -4.], [2e-5, -1e-5, 0.]]) * rng.uniform(-1, 1, (3, 3))
        labels[name] = [np.linalg.inv(Hf), Hf]
        truth[name] = (imgs[0], imgs[1], mask, Hf)
    np.save(os.path.join(root, 'BasesHomo_small.npy'), labels, allow_pickle=True)
    return truth


@pytest.mark.parametrize('size', [128, 64])
def test_unhomo_train_data_vs_oracle(tmp_path, size):
    from dmhomo_amd.dataset import UnHomoTrainData, ConditionLoader
    truth = _make_dataset(str(tmp_path))
    ds = UnHomoTrainData(str(tmp_path), size, device=dev())
    assert len(ds) == 5
    data, cls = ds.batch(list(range(5)))
    assert data.shape == (5, 12, size, size) and cls.dtype == torch.long and int(cls.abs().sum()) == 0
    got = data.cpu().numpy()
    for i, name in enumerate(ds.im1_im2_names):
        want = OD.build_item(*truth[name], size)
        err_img = np.abs(got[i, :6] - want[:6]).max()
        assert err_img < 1e-6, (name, err_img)                       # bilinear resize of both frames
        assert np.array_equal(got[i, 6], want[6]), name              # nearest + opening: exact
        assert np.abs(got[i, 7:10] - want[7:10]).max() < 2e-5, name  # HSV flow image (pinned elsewhere, G3)
        assert np.abs(got[i, 10:] - want[10:]).max() < 1e-5, name    # flow (G2)
    item, c = ds[3]
    assert c == 0 and torch.equal(item, data[3])
    # the loader: a permutation per epoch, short last batch, endless
    dl = ConditionLoader(ds, 2, shuffle=True, seed=1)
    sizes = [next(dl)[0].shape[0] for _ in range(6)]
    assert sizes == [2, 2, 1, 2, 2, 1]


def test_unhomo_train_data_mixed_source_sizes(tmp_path):
    """frames / masks of different sizes in one batch (per-item uploads instead of one stacked copy)"""
    from dmhomo_amd.dataset import UnHomoTrainData
    truth = _make_dataset(str(tmp_path), n=4, ragged=True)
    ds = UnHomoTrainData(str(tmp_path), 96, device=dev())
    got = ds.batch([0, 1, 2, 3])[0].cpu().numpy()
    for i, name in enumerate(ds.im1_im2_names):
        want = OD.build_item(*truth[name], 96)
        assert np.abs(got[i, :6] - want[:6]).max() < 1e-6 and np.array_equal(got[i, 6], want[6]), name


def test_trainer_trains_from_dataset_folder(tmp_path):
    """Trainer(folder=<dataset dir>) wires UnHomoTrainData + the loader into Trainer.train (DDP:1735-1752)"""
    from test_gpu_unet import make_cfg
    from dmhomo_amd import cfg, ddpm
    _make_dataset(str(tmp_path / 'data'), n=4)
    m, _ = make_cfg(8)
    d = cfg.GaussianDiffusion(m, image_size=32, timesteps=1000, sampling_timesteps=2, objective='pred_x0').to(dev())
    tr = ddpm.Trainer(d, str(tmp_path / 'data'), train_batch_size=2, train_lr=1e-3, train_num_steps=2,
                      results_folder=str(tmp_path / 'res'), num_worker=2)
    losses = []
    tr.train(log=lambda s, l: losses.append(float(l)))
    assert len(losses) == 2 and all(np.isfinite(l) for l in losses)

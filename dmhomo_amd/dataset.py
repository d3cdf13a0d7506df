"""Condition dataset path: ``UnHomoTrainData`` (DDP:1045-1163) re-hosted for the MI355X.

The reference builds every item in DataLoader worker processes with OpenCV (PNG decode, two bilinear resizes, a nearest
resize + 3x3 opening of the mask, the homography flow and its HSV image in numpy) and ships float32 items to the GPU.
Here the host only *decodes* (PIL; worker threads) and hands uint8 pixels over; everything else is built on the
device a batch at a time, straight into the 12-channel batch tensor [img1(3) img2(3) mask(1) rgb_flow(3) flow(2)]:

    dmh_resize_bilinear_u8 x2, dmh_mask_open_nearest, dmh_homography_flow      (csrc/dataset.hip, csrc/geometry.hip)

so the H2D copy carries 1 byte per source pixel-channel instead of 4 bytes per item element, and no CPU core touches a
pixel after decoding.  File layout (the reference hard-codes ``/root/test/trainset/Contant-Aware-DeepH-Data/Data/Train``,
DDP:1058-1067; here it is ``benchmark_path`` when that directory holds the label file):

    <root>/BasesHomo_small.npy                    dict  "<dir>_<a>_<dir>_<b>" -> [homo_b, homo_f, ...]   (pickled)
    <root>/HomoGAN_Bug_Masks/<pair name>.npy      mask, any shape that squeezes to HxW
    <root>/<dir>/<dir>_<a>.png, <dir>_<b>.png     frames
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops
from .ddpm import adapt_homography_to_preprocessing_v3, homo_to_flow_rgb

REFERENCE_ROOT = '/root/test/trainset/Contant-Aware-DeepH-Data/Data/Train'      # DDP:1058


def _imread_bgr(path):
    """cv2.imread(path): HxWx3 uint8, BGR channel order (alpha dropped, grey replicated)."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


class UnHomoTrainData:
    """DDP:1045-1163.  ``ds[i]`` -> ((12, S, S) float32 tensor on ``device``, 0) like the reference's items;
    ``ds.batch(indices)`` builds a whole batch with one launch per stage (what ``ConditionLoader`` uses)."""

    def __init__(self, benchmark_path, image_size, exts=('jpg', 'jpeg', 'png', 'tiff'), augment_horizontal_flip=False,
                 convert_image_to=None, device=None, workers=8):
        root = benchmark_path if benchmark_path and os.path.isfile(os.path.join(str(benchmark_path), 'BasesHomo_small.npy')) \
            else REFERENCE_ROOT
        self.trainset_pth = str(root)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.pseudo_labels = np.load(os.path.join(self.trainset_pth, 'BasesHomo_small.npy'), allow_pickle=True).item()
        self.im1_im2_names = list(self.pseudo_labels.keys())
        self.image_size = image_size
        self._pool = ThreadPoolExecutor(max_workers=workers)

    def __len__(self):
        return len(self.im1_im2_names)

    # ---- host side of one item: names (DDP:1104-1108), label (DDP:1110-1113), mask file (DDP:1116), the two PNGs
    def _load(self, idx):
        name = self.im1_im2_names[idx]
        parts = name.split('_')
        dir_name = parts[0]
        im1_name, im2_name = '_'.join(parts[:2]) + '.png', '_'.join(parts[2:]) + '.png'
        homo_f = np.asarray(self.pseudo_labels[name][1], dtype=np.float64).reshape(3, 3)
        mask = np.load(os.path.join(self.trainset_pth, 'HomoGAN_Bug_Masks', name + '.npy'))
        img1 = _imread_bgr(os.path.join(self.trainset_pth, dir_name, im1_name))
        img2 = _imread_bgr(os.path.join(self.trainset_pth, dir_name, im2_name))
        homo = adapt_homography_to_preprocessing_v3(360, 640, homo_f, self.image_size, self.image_size)   # DDP:1138
        return img1, img2, np.squeeze(mask).astype(np.float32), homo

    def decode_async(self, indices):
        """start loading / decoding the files of ``indices`` on the worker threads -> futures for ``assemble``"""
        return [self._pool.submit(self._load, i) for i in indices]

    def batch(self, indices):
        return self.assemble([f.result() for f in self.decode_async(indices)])

    def assemble(self, items):
        """decoded items -> ((B, 12, S, S) batch on the device, classes): everything after the decode, on the GPU"""
        S, dev = self.image_size, self.device
        B = len(items)
        out = torch.empty((B, 12, S, S), device=dev, dtype=torch.float32)

        def stacked(k, dtype):
            shapes = {it[k].shape for it in items}
            if len(shapes) == 1:
                return [(torch.from_numpy(np.stack([it[k] for it in items])).to(dev, non_blocking=True), 0, B)]
            return [(torch.from_numpy(np.ascontiguousarray(it[k]))[None].to(dev), i, 1) for i, it in enumerate(items)]

        bstride = 12 * S * S

        def plane(b0, c0):                # address of channel plane c0 of image b0 inside the batch tensor
            return ops.C.c_void_p(out.data_ptr() + 4 * (b0 * bstride + c0 * S * S))

        for k, c0 in ((0, 0), (1, 3)):
            for src, b0, nb in stacked(k, np.uint8):
                _, hs, ws, ch = src.shape
                ops.call('dmh_resize_bilinear_u8', ops.C.c_void_p(src.data_ptr()), plane(b0, c0), nb, hs, ws, ch, S, S,
                         bstride, 255.)
        for src, b0, nb in stacked(2, np.float32):
            _, hs, ws = src.shape
            ops.call('dmh_mask_open_nearest', ops.ptr(src), plane(b0, 6), nb, hs, ws, S, S, bstride)
        flow, rgb = homo_to_flow_rgb(np.stack([it[3] for it in items]), S, S)         # DDP:1157-1160
        out[:, 7:10] = rgb
        out[:, 10:12] = flow
        return out, torch.zeros((B,), dtype=torch.long, device=dev)                  # scene_class = 0, DDP:1136

    def __getitem__(self, idx):
        data, cls = self.batch([idx])
        return data[0], 0


class ConditionLoader:
    """endless batches like ``cycle(DataLoader(ds, batch_size, shuffle))`` (DDP:1746-1752): a new permutation per epoch, the
    last short batch kept; the next batch's files are decoded by the worker threads while the current one is in use."""

    def __init__(self, ds, batch_size, shuffle=True, seed=0, rank=0, world=1):
        self.ds, self.batch_size, self.shuffle = ds, batch_size, shuffle
        self.gen = torch.Generator().manual_seed(seed)
        self.rank, self.world = rank, world
        self._order, self._pos = [], 0
        self._pending = None

    def _next_indices(self):
        if self._pos >= len(self._order):
            n = len(self.ds)
            order = torch.randperm(n, generator=self.gen).tolist() if self.shuffle else list(range(n))
            self._order, self._pos = order[self.rank::self.world], 0
        idx = self._order[self._pos:self._pos + self.batch_size]
        self._pos += self.batch_size
        return idx

    def __iter__(self):
        return self

    def __next__(self):
        cur = self._pending if self._pending is not None else self.ds.decode_async(self._next_indices())
        self._pending = self.ds.decode_async(self._next_indices())     # decoded while the caller trains on ``cur``
        return self.ds.assemble([f.result() for f in cur])

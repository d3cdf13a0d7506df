#!/usr/bin/env python3
"""Time the condition dataset path on a synthetic copy of the CA-Homo layout (640x360 PNG pairs):
items/s of UnHomoTrainData.batch (PNG decode on host threads + device preprocessing), the device part alone, and the
oracle's numpy restatement of the reference's per-item OpenCV pipeline on one host core."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    from test_gpu_dataset import _make_dataset
    from dmhomo_amd.dataset import UnHomoTrainData
    from oracle import dataset as OD
    bs, size = 16, 128
    with tempfile.TemporaryDirectory() as tmp:
        truth = _make_dataset(tmp, n=32)
        ds = UnHomoTrainData(tmp, size, device=torch.device('cuda', 0), workers=8)
        idx = list(range(32))
        ds.batch(idx[:bs])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(6):
            ds.batch(idx[(r % 2) * bs:(r % 2) * bs + bs])
        torch.cuda.synchronize()
        full = (time.perf_counter() - t0) / 6
        # device part alone: the same launches on pre-decoded pixels
        items = [ds._load(i) for i in idx[:bs]]
        ds.assemble(items)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ds.assemble(items)
        torch.cuda.synchronize()
        devpart = (time.perf_counter() - t0) / 10
        name = ds.im1_im2_names[0]
        t0 = time.perf_counter()
        for _ in range(4):
            OD.build_item(*truth[name], size)
        cpu = (time.perf_counter() - t0) / 4
    print(json.dumps({'batch': bs, 'source': '640x360 PNG pairs', 'image_size': size,
                      'items_per_s_with_decode': bs / full, 'ms_per_batch_with_decode': full * 1e3,
                      'ms_per_batch_after_decode': devpart * 1e3,
                      'oracle_numpy_ms_per_item_1core': cpu * 1e3}))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Counterpart of the reference's DGM/generate_nyps_to_single_case.py (GEN:22-50): split the list-of-dict record
files written by scripts/dgm_sample.py ({"imgs": uint8 (B,6,H,W), "homos": float64 (B,3,3)} per entry) into one
file per sample, {"img12": (6,H,W) uint8, "homo12": (3,3) float64}, the format HEM/dataset/data_loader.py:123-131
reads with np.load(...).item().

    python scripts/generate_nyps_to_single_case.py [--src 'traindata/test/dataset/*npy*'] [--dst traindata/samples]

The reference's visual unit_test (cv2.warpPerspective + GIF, GEN:8-19) needs cv2 / imageio and is not reproduced; the
numeric check it stands for — img1 warped by homo12 lands on img2 — is covered by tests/test_gpu_geometry.py.
Sample numbering starts at 1 and continues across files, as in the reference (GEN:44-47)."""
import argparse
import glob
import os

import numpy as np


def split_records(paths, dst, start_idx=0, verbose=True):
    """-> number of samples written so far (the reference's running ``idx``)."""
    os.makedirs(dst, exist_ok=True)
    idx = start_idx
    for npy in paths:
        buf = np.load(npy, allow_pickle=True)
        if verbose:
            print(f'process {npy}\nit contains {len(buf)} samples')
        for item in buf:
            imgs, homos = item['imgs'], item['homos']
            if verbose:
                print(f'imgs shape {imgs.shape} | homos shape {homos.shape}')
            assert len(imgs) == len(homos), (imgs.shape, homos.shape)
            for i in range(len(imgs)):
                idx += 1
                np.save(os.path.join(dst, f'{idx}.npy'), {'img12': imgs[i], 'homo12': homos[i]})
    return idx


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--src', default='traindata/test/dataset/*npy*')        # GEN:23
    ap.add_argument('--dst', default='traindata/samples')                   # GEN:47
    a = ap.parse_args()
    n = split_records(sorted(glob.glob(a.src)), a.dst)
    print(f'{n} samples -> {a.dst}')

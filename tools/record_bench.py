"""measurement of SURVEY 8f row 2, the post-sample record path (saveTrainPair: uint8 export + DLT homography):
dmhomo_amd.ddpm.saveTrainPair on the GPU vs the oracle's restatement of the reference (pinv of the 2HW x 8 system per
sample, DDP:1577-1661) on the host cores.  Test/bench infrastructure: imports the oracle as the CPU baseline only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dmhomo_amd import ddpm
from oracle import geometry as OG

dev = torch.device('cuda', 0)
for S, B in ((128, 25), (256, 25)):
    g = torch.Generator().manual_seed(S)
    homos = np.stack([np.eye(3) + 0.02 * torch.randn(3, 3, generator=g, dtype=torch.float64).numpy() * [[1, 1, 100], [1, 1, 100], [1e-3, 1e-3, 0]] for _ in range(B)])
    flow, _ = ddpm.homo_to_flow_rgb(homos, S, S)                      # (B,2,S,S) on the GPU
    imgs = torch.rand(B, 6, S, S, device=dev)
    mask = torch.ones(B, 1, S, S, device=dev)
    for _ in range(2):
        rec = ddpm.saveTrainPair(imgs, mask, flow)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        rec = ddpm.saveTrainPair(imgs, mask, flow)                    # includes the device -> host copies of the record
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / 5 * 1e3
    fc = flow[:2].cpu()
    t0 = time.perf_counter()
    ref = OG.homo_gen(fc)
    cpu_s = (time.perf_counter() - t0) / 2
    err = np.abs(rec['homos'][:2] - ref.numpy().reshape(2, 3, 3)).max()
    print(f'{S}x{S}, {B} samples: saveTrainPair {gpu_ms:.2f} ms per batch on the GPU ({gpu_ms / B:.3f} ms / sample); '
          f'reference algorithm on the CPU {cpu_s * 1e3:.0f} ms / sample ({torch.get_num_threads()} threads); '
          f'max |H_gpu - H_pinv| = {err:.2e}')

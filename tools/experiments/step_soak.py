"""soak of the WHOLE headline step (round 4): the same keyed 25-image sample() N times through the replayed graph ('streams'
mode, both passes concurrent) — every call bitwise the first.  A timing-dependent fault anywhere in the ~330 kernels of a
denoise step (an LDS slot reused a barrier early, a register read before an MFMA wrote it) shows up as a differing call.
    python tools/experiments/step_soak.py [N=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmhomo_amd import cfg, ddpm
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
model = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
model.cfg_mode = 'streams'
d = cfg.GaussianDiffusion(model, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev)
d.hip_graph = True
data, classes = next(ddpm.SyntheticConditions(128, 25, seed=1000, device=dev))
rgb_flow, flow, mask = data[:, -5:-2].contiguous(), data[:, -2:].contiguous(), data[:, -6:-5].contiguous()
def run():
    d.rng.key_by_sample(99, range(25), dev)
    return d.sample(classes, rgb_flow, flow, mask)[0].clone()
first = run()
bad, t0 = 0, time.perf_counter()
for i in range(N):
    bad += int(not torch.equal(run(), first))
torch.cuda.synchronize()
print(f'{N} sample() calls of 25 images x 32 steps in {time.perf_counter() - t0:.1f} s: {bad} differ from the first')
sys.exit(1 if bad else 0)

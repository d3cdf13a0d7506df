"""how long the host spends inside the HIP-graph replay of one sampling call, and where the GPU idles between calls
(round 2): gpurun -- python tools/experiments/replay_timing.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmhomo_amd import cfg, ddpm, ops

dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
model.cfg_mode = 'streams'
diffusion = cfg.GaussianDiffusion(model, image_size=128, timesteps=1000, sampling_timesteps=32, loss_type='l1',
                                  objective='pred_x0').to(dev)
diffusion.hip_graph = True
conds = ddpm.SyntheticConditions(128, 25, seed=1000, device=dev)
data, classes = next(conds)
rgb_flow, flow, mask = data[:, -5:-2].contiguous(), data[:, -2:].contiguous(), data[:, -6:-5].contiguous()
for _ in range(2):
    diffusion.sample(classes, rgb_flow, flow, mask)
torch.cuda.synchronize()
g = diffusion.__dict__['_graph_state']['graph']
orig = g.replay
spent = []
def timed():
    t = time.perf_counter(); orig(); spent.append(time.perf_counter() - t)
g.replay = timed
t0 = time.perf_counter()
marks = []
for _ in range(4):
    ta = time.perf_counter()
    img, _, fl = diffusion.sample(classes, rgb_flow, flow, mask)
    tb = time.perf_counter()
    u8 = ops.to_uint8(img); h = ops.dlt_homography(fl)
    tc = time.perf_counter()
    marks.append((tb - ta, tc - tb))
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print('4 calls: %.1f ms each; host time inside graph.replay(): %s ms; sample() host time %s ms; post-processing enqueue %s ms'
      % (tot / 4 * 1e3, [round(s * 1e3, 1) for s in spent], [round(a * 1e3, 1) for a, b in marks], [round(b * 1e3, 2) for a, b in marks]))

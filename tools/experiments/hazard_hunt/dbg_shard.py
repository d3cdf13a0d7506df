import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import dev
from dmhomo_amd import cfg, ops
from test_gpu_unet import make_cfg
from test_gpu_rng import _fullsize_inputs
m, _ = make_cfg(64)
m.cfg_mode = 'streams'
d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=int(os.environ.get('S', '32')), objective='pred_x0').to(dev())
d.hip_graph = os.environ.get('GRAPH', '1') == '1'
ins = [t.to(dev()) for t in _fullsize_inputs(25)]
def run(lo, hi):
    d.rng.key_by_sample(7, range(lo, hi), dev())
    rf01, flow, mk, c = (t[lo:hi].contiguous() for t in ins)
    return d.sample(c, rf01, flow, mk)[0].clone()
whole = run(0, 25)
whole2 = run(0, 25)
print('whole repeat equal', torch.equal(whole, whole2))
s0, s1 = run(0, 13), run(13, 25)
cat = torch.cat([s0, s1])
for r in range(25):
    e = float((cat[r] - whole[r]).abs().max())
    if e: print('row', r, 'maxdiff', e)
print('equal', torch.equal(cat, whole))

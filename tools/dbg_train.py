"""dev tool: stage-by-stage comparison of UnetTrain.forward (training path, activations saved) with the inference engine"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from test_gpu_unet import make_cfg, _cond_inputs, g
from dmhomo_amd import train
m, sd = make_cfg(8)
B, S = 3, 16
x, rf, mk = _cond_inputs(B, S, 700)
t = torch.tensor([17, 503, 998]); c = torch.zeros(B, dtype=torch.long); keep = torch.tensor([True, False, True])
ut = train.UnetTrain(m)
tt, te = {}, {}
out, saved = ut.forward(g(x), g(t), g(c), g(rf), g(mk), g(keep), taps=tt)
oe = m._run(g(x), g(t), g(c), g(rf), g(mk), [g(keep).to(torch.uint8)], taps=te)
for k in te:
    if k in tt:
        a, b = tt[k].double(), te[k].double()
        print(f'{k:18s} rel {((a - b).abs().max() / b.abs().max()).item():.3e}')
print('out rel', ((out - oe).abs().max() / oe.abs().max()).item())

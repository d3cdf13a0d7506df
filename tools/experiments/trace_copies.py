"""dev tool: where one optimiser step of dmhomo_amd.train makes torch copy a tensor (non-contiguous .contiguous(), copy_, clone,
zero_) — call site and shape, by count.   python tools/experiments/trace_copies.py"""
import collections, os, sys, traceback
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/dmhomo_amd') else os.environ.get('GRAFT_REPO_ROOT','.'))
import torch
from dmhomo_amd import cfg, train
from dmhomo_amd.ddpm import SyntheticConditions
dev = torch.device('cuda', 0)
torch.manual_seed(1234)
m = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1).to(dev)
d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0').to(dev)
ts = train.TrainStep(d, lr=5e-4, betas=(0.9, 0.99), accum=1)
img, cls = next(SyntheticConditions(128, 16, seed=1000, device=dev))
img[:, :6] = torch.rand((16, 6, 128, 128), device=dev)
for _ in range(2):
    ts.step([(img, cls)])
torch.cuda.synchronize()
sites = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'dmhomo_amd' in fr.filename:
            return f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.line[:70]}'
    return '?'
oc, ocp, ocl, oz, oto = torch.Tensor.contiguous, torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.zero_, torch.Tensor.to
def c(self, *a, **k):
    if not self.is_contiguous(): sites['contiguous ' + str(tuple(self.shape)) + ' ' + site()] += 1
    return oc(self, *a, **k)
def cp(self, *a, **k):
    sites['copy_ ' + str(tuple(self.shape)) + ' ' + site()] += 1
    return ocp(self, *a, **k)
def cl(self, *a, **k):
    sites['clone ' + str(tuple(self.shape)) + ' ' + site()] += 1
    return ocl(self, *a, **k)
def z(self, *a, **k):
    sites['zero_ ' + str(tuple(self.shape)) + ' ' + site()] += 1
    return oz(self, *a, **k)
torch.Tensor.contiguous, torch.Tensor.copy_, torch.Tensor.clone, torch.Tensor.zero_ = c, cp, cl, z
ts.step([(img, cls)])
torch.cuda.synchronize()
for s, n in sites.most_common(60):
    print(n, s)
print(sum(sites.values()))

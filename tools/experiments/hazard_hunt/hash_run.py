import sys, hashlib, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from dmhomo_amd import cfg, ddpm
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
m.cfg_mode = 'streams'
d = cfg.GaussianDiffusion(m, image_size=128, timesteps=1000, sampling_timesteps=4, objective='pred_x0').to(dev)
conds = ddpm.SyntheticConditions(128, 3, seed=1000, device=dev)
data, classes = next(conds)
rgb_flow, flow, mask = data[:, -5:-2].contiguous(), data[:, -2:].contiguous(), data[:, -6:-5].contiguous()
torch.manual_seed(5)
img, _, _ = d.sample(classes, rgb_flow, flow, mask)
print('HASH', hashlib.sha256(img.cpu().numpy().tobytes()).hexdigest()[:16], float(img.double().sum()))

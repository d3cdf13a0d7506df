"""probe: two independent half-batch denoising chains (each cond + null rows in one launch sequence) on two HIP streams,
the second one started half a forward late, against the product's two-pass 'streams' mode.  Eager launches on both sides.
    python tools/experiments/chains_probe.py [skew_rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmhomo_amd import cfg, ddpm, ops
from dmhomo_amd.cfg import ddim_pairs

dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
diffusion = cfg.GaussianDiffusion(model, image_size=128, timesteps=1000, sampling_timesteps=32, loss_type='l1',
                                  objective='pred_x0').to(dev)
B = 25
conds = ddpm.SyntheticConditions(128, B, seed=1000, device=dev)
data, classes = next(conds)
rgb_flow, flow, mask = data[:, -5:-2].contiguous(), data[:, -2:].contiguous(), data[:, -6:-5].contiguous()
skew_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 6


def baseline():
    model.cfg_mode = 'streams'
    diffusion.sample(classes, rgb_flow, flow, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        diffusion.sample(classes, rgb_flow, flow, mask)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 2


def chains(skew):
    model.cfg_mode = 'batched'
    host = diffusion._host()
    rf = ops.affine(rgb_flow.to(torch.float32), 2., -1.)
    bounds = [(0, 13), (13, 25)]
    streams = [torch.cuda.Stream(device=dev) for _ in bounds]
    cur = torch.cuda.current_stream()
    imgs = []
    for (lo, hi), st in zip(bounds, streams):
        st.wait_stream(cur)
        imgs.append(torch.randn((hi - lo, 6, 128, 128), device=dev))
    if skew:
        with torch.cuda.stream(streams[1]):          # a head start for chain 0 = a dummy partial forward in front of chain 1
            tc = torch.full((skew,), 999, device=dev, dtype=torch.long)
            diffusion._network(imgs[1][:skew].contiguous(), tc, classes[:skew], rf[:skew].contiguous(), mask[:skew].contiguous(), 3.)
    for time_, time_next in ddim_pairs(diffusion.num_timesteps, diffusion.sampling_timesteps):
        for ci, ((lo, hi), st) in enumerate(zip(bounds, streams)):
            with torch.cuda.stream(st):
                b = hi - lo
                tc = torch.full((b,), time_, device=dev, dtype=torch.long)
                cond, null = diffusion._network(imgs[ci], tc, classes[lo:hi], rf[lo:hi].contiguous(), mask[lo:hi].contiguous(), 3.)
                if time_next < 0:
                    step = diffusion._step(host, time_, ops.MODE_LAST, 3., True)
                    noise = None
                else:
                    step = diffusion._step(host, time_, ops.MODE_DDIM, 3., True, diffusion._ddim_coef(host, time_, time_next))
                    noise = torch.randn_like(imgs[ci])
                imgs[ci], _, _ = ops.sampler_step(step, cond, null, imgs[ci], noise, want_x_start=False)
    for st in streams:
        cur.wait_stream(st)
    return imgs


for rep in range(2):
    tb = baseline()
    chains(skew_rows)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    chains(skew_rows)
    torch.cuda.synchronize()
    tc_ = time.perf_counter() - t0
    chains(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    chains(0)
    torch.cuda.synchronize()
    t0_ = time.perf_counter() - t0
    print(f'streams (2 passes x 25 rows): {B / tb:6.2f} img/s | chains 13+12 skew {skew_rows} rows: {B / tc_:6.2f} img/s | chains no skew: {B / t0_:6.2f} img/s', flush=True)

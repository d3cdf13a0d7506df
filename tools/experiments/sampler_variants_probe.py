"""probe (round 4): GaussianDiffusion constructor / call arguments away from the DGM's values, product vs oracle on
replayed draws: eta 0 / 0.5, linear schedule, T = 200, cond_scale 1 / 0.5 / 7, cond_drop_prob 0 / 1 / 0.2, graph on / off"""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import dev, ReplayDeviceRng
from detweights import det_state_dict, shapes_of
from oracle import diffusion as OD
from dmhomo_amd import cfg
cases = [
    dict(eta=0.0), dict(eta=0.5), dict(schedule='linear'), dict(T=200, S=7), dict(cond_scale=1.0), dict(cond_scale=0.5),
    dict(cond_scale=7.0), dict(drop=0.0), dict(drop=1.0), dict(drop=0.2), dict(objective='pred_v', eta=0.3, schedule='linear'),
    dict(S=1), dict(T=50, S=50 - 1),
]
for kw in cases:
    for graph in (False, True):
        try:
            T, S = kw.get('T', 1000), kw.get('S', 5)
            m = cfg.Unet(dim=8, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1, cond_drop_prob=kw.get('drop', 0.5))
            sd = det_state_dict(shapes_of(m), 0)
            m.load_state_dict(sd)
            m = m.to(dev())
            d = cfg.GaussianDiffusion(m, image_size=16, timesteps=T, sampling_timesteps=S, objective=kw.get('objective', 'pred_x0'),
                                      beta_schedule=kw.get('schedule', 'cosine'), ddim_sampling_eta=kw.get('eta', 1.)).to(dev())
            d.hip_graph = graph
            B = 2
            g = torch.Generator().manual_seed(9)
            rf01 = torch.rand(B, 3, 16, 16, generator=g)
            mk = (torch.rand(B, 1, 16, 16, generator=g) > 0.4).float()
            fl = torch.randn(B, 2, 16, 16, generator=g)
            c = torch.zeros(B, dtype=torch.long)
            torch.manual_seed(4)
            rec = OD.RecordRng()
            with torch.no_grad():
                ref, _, _ = OD.cfg_sample(sd, OD.schedule_buffers(T, kw.get('schedule', 'cosine')), c, rf01, fl, mk, image_size=16, channels=6,
                                          sampling_timesteps=S, objective=kw.get('objective', 'pred_x0'), cond_scale=kw.get('cond_scale', 3.),
                                          cond_drop_prob=kw.get('drop', 0.5), eta=kw.get('eta', 1.), rng=rec)
            if graph:
                # the graph path needs the stock generator: compare graph against eager on the device stream instead
                torch.manual_seed(11)
                d.hip_graph = False
                a = d.sample(c.to(dev()), rf01.to(dev()), fl.to(dev()), mk.to(dev()), cond_scale=kw.get('cond_scale', 3.))[0].clone()
                torch.manual_seed(11)
                d.hip_graph = True
                b = d.sample(c.to(dev()), rf01.to(dev()), fl.to(dev()), mk.to(dev()), cond_scale=kw.get('cond_scale', 3.))[0].clone()
                print('OK  ' if torch.equal(a, b) else 'DIFF', 'graph==eager', kw)
                continue
            d.rng = ReplayDeviceRng(rec.draws)
            img, _, _ = d.sample(c.to(dev()), rf01.to(dev()), fl.to(dev()), mk.to(dev()), cond_scale=kw.get('cond_scale', 3.))
            err = float((img.cpu() - ref).abs().max())
            fin = bool(torch.isfinite(ref).all())
            print('OK  ' if (err < 4e-4 or not fin) else 'BAD ', f'vs oracle err={err:.2e} ref_finite={fin} draws={len(rec.draws)} used={d.rng.i}', kw)
        except Exception as e:
            print('FAIL', kw, 'graph' if graph else 'eager', type(e).__name__, str(e)[:200])

"""CPU: host-side logic of the reference-shaped classes (no oracle, no GPU): state_dict layout,
schedule buffers against the reference's golden values, DDIM time grid, constructor contracts."""
import json
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope='module')
def meta(golden_dir):
    with open(os.path.join(golden_dir, 'meta.json')) as f:
        return json.load(f)


def shapes(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def test_state_dict_layout_matches_reference(meta):
    from dmhomo_amd import cfg, ddpm
    assert shapes(cfg.Unet(dim=8, channels=6, num_classes=1)) == meta['unet_cfg_tiny_keys']
    m64 = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    assert shapes(m64) == meta['unet_cfg_dim64_keys']
    assert sum(p.numel() for p in m64.parameters()) == 38417734
    d = cfg.GaussianDiffusion(m64, image_size=128, timesteps=1000, sampling_timesteps=32, objective='pred_x0')
    assert shapes(d) == meta['diffusion_cfg_dim64_keys']
    assert shapes(ddpm.Unet(dim=8, channels=3)) == meta['unet_ddp_tiny_nosc_keys']
    assert shapes(ddpm.Unet(dim=8, channels=3, self_condition=True)) == meta['unet_ddp_tiny_sc_keys']
    assert shapes(ddpm.Unet(dim=64, channels=3)) == meta['unet_ddp_dim64_keys']


def test_import_paths_of_the_reference_scripts():
    from dmhomo_amd.denoising_diffusion_models.denoising_diffusion_pytorch import Trainer  # noqa: F401
    from dmhomo_amd.denoising_diffusion_models.classifier_free_guidance import Unet, GaussianDiffusion  # noqa: F401


def test_schedule_buffers_match_reference(golden_dir):
    from dmhomo_amd import cfg, ddpm
    from dmhomo_amd.schedule import ddim_pairs
    g = np.load(os.path.join(golden_dir, 'schedule.npz'))
    tiny = cfg.Unet(dim=8, channels=6, num_classes=1)
    for sched, T in (('cosine', 1000), ('linear', 1000), ('cosine', 10)):
        d = cfg.GaussianDiffusion(tiny, image_size=16, timesteps=T, beta_schedule=sched)
        names = [k for k in d.state_dict() if not k.startswith('model.')]
        assert len(names) == 13
        for k in names:
            v = d.state_dict()[k]
            assert v.dtype == torch.float32 and torch.equal(v, torch.from_numpy(g[f'{sched}{T}.{k}'])), (sched, k)
    d2 = ddpm.GaussianDiffusion(ddpm.Unet(dim=8, channels=3), image_size=16, timesteps=1000)
    assert torch.equal(d2.alphas_cumprod, torch.from_numpy(g['cosine1000.alphas_cumprod']))
    for S in (4, 32, 250):
        pairs = ddim_pairs(1000, S)
        assert [p[0] for p in pairs] + [pairs[-1][1]] == g[f'times{S}'].tolist()
    assert [p[0] for p in ddim_pairs(1000, 1000)] == list(range(999, -1, -1))       # S == T grid


def test_constructor_contracts():
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=8, channels=6, num_classes=1)
    assert (m.channels, m.out_dim, m.cond_drop_prob, m.random_or_learned_sinusoidal_cond) == (6, 6, 0.5, False)
    with pytest.raises(ValueError):
        cfg.GaussianDiffusion(m, image_size=16, beta_schedule='nope')
    with pytest.raises(AssertionError):
        cfg.GaussianDiffusion(m, image_size=16, objective='bad')
    with pytest.raises(AssertionError):
        cfg.GaussianDiffusion(cfg.Unet(dim=8, channels=6, num_classes=1, out_dim=3), image_size=16)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=32)
    assert d.is_ddim_sampling and d.num_timesteps == 1000 and d.sampling_timesteps == 32
    assert not cfg.GaussianDiffusion(m, image_size=16).is_ddim_sampling
    with pytest.raises(TypeError):       # the reference's ancestral path is uncallable
        cfg.GaussianDiffusion(m, image_size=16).sample(torch.zeros(1, dtype=torch.long), None, None, None)


def test_no_cpu_path():
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=8, channels=6, num_classes=1)
    x = torch.zeros(1, 6, 16, 16)
    with pytest.raises(RuntimeError, match='GPU'):
        m(x, torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long), torch.zeros(1, 3, 16, 16),
          torch.zeros(1, 1, 16, 16))


def test_product_does_not_import_the_oracle():
    import subprocess
    import sys
    code = 'import sys, dmhomo_amd, dmhomo_amd.distributed; print(any(m == "oracle" or m.startswith("oracle.") for m in sys.modules))'
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip() == 'False', out.stdout + out.stderr


def test_record_files_split_into_the_hem_sample_format(tmp_path):
    """row 2 of SURVEY 8f: dgm_sample's list-of-dict record (SAMPLE:73-77) -> one {"img12", "homo12"} file per sample
    (GEN:36-47), readable the way HEM/dataset/data_loader.py:123-131 reads it"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'gen_split', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts',
                                  'generate_nyps_to_single_case.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rng = np.random.default_rng(0)
    recs = [{'imgs': rng.integers(0, 256, (3, 6, 8, 8), dtype=np.uint8), 'homos': rng.standard_normal((3, 3, 3))}
            for _ in range(2)]
    src = tmp_path / 'dataset'
    src.mkdir()
    np.save(str(src / 'idx_0_rank_0_part_0_dm_cahomo_0.006k.npy'), recs)
    n = gen.split_records(sorted(str(p) for p in src.glob('*npy*')), str(tmp_path / 'samples'), verbose=False)
    assert n == 6
    for k in range(1, 7):
        buf = np.load(str(tmp_path / 'samples' / f'{k}.npy'), allow_pickle=True).item()      # as the HEM loader does
        r, i = divmod(k - 1, 3)
        assert buf['img12'].dtype == np.uint8 and np.array_equal(buf['img12'], recs[r]['imgs'][i])
        assert buf['homo12'].dtype == np.float64 and np.array_equal(buf['homo12'], recs[r]['homos'][i])


def test_ema_decay_schedule():
    """EMA stand-in: ema_pytorch's published warm-up — 0 until update_after_step, then
    clamp(1 - (1 + epoch / inv_gamma) ** -power, min_value, beta) with epoch = step - update_after_step - 1"""
    import torch
    from dmhomo_amd.ddpm import EMA
    e = EMA(torch.nn.Linear(2, 2), beta=0.995, update_every=10)
    assert (e.update_after_step, e.inv_gamma, abs(e.power - 2 / 3) < 1e-12, e.min_value) == (100, 1.0, True, 0.0)
    for step, want in ((0, 0.0), (100, 0.0), (101, 0.0), (102, 1 - 2 ** (-2 / 3)), (111, 1 - 11 ** (-2 / 3)),
                       (100000, 0.995)):
        e.step.fill_(step)
        assert abs(e.get_current_decay() - want) < 1e-12, (step, e.get_current_decay(), want)
    sd = e.state_dict()
    assert {'initted', 'step', 'online_model.weight', 'ema_model.weight'} <= set(sd)

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


def _ema_shaped_checkpoint(d_online_sd, d_ema_sd, with_online=True, step=1234):
    """a checkpoint dict shaped like the reference's Trainer.save output (DDP:1786-1802) whose 'ema' entry has the key
    layout of ema_pytorch.EMA(diffusion_model).state_dict(): 'initted', 'step', 'ema_model.<key>' and —
    unless the EMA was built with include_online_model=False or the file was stripped — 'online_model.<key>', where
    <key> runs over GaussianDiffusion.state_dict() ('model.<unet key>' + the 13 schedule buffers)."""
    ema = {'initted': torch.tensor(True), 'step': torch.tensor(step)}
    if with_online:
        ema.update({'online_model.' + k: v.clone() for k, v in d_online_sd.items()})
    ema.update({'ema_model.' + k: v.clone() for k, v in d_ema_sd.items()})
    return {'step': step, 'model': {k: v.clone() for k, v in d_online_sd.items()}, 'opt': None, 'ema': ema,
            'scaler': None, 'version': '1.0.0'}


@pytest.mark.parametrize('with_online', [True, False])
def test_trainer_load_of_an_ema_pytorch_shaped_checkpoint(tmp_path, meta, with_online):
    """SURVEY §8f row 3 / DDP:1804-1826: Trainer.load of a checkpoint whose EMA weights differ from the online ones.
    The online model must end up with data['model'], ``trainer.ema.ema_model`` (what Trainer.sample reads, DDP:1960)
    with the 'ema_model.' weights, in storage of its own — also when the file carries no 'online_model.' keys."""
    from detweights import det_state_dict, shapes_of
    from dmhomo_amd import cfg, ddpm

    def build(seed):
        m = cfg.Unet(dim=8, channels=6, num_classes=1)
        m.load_state_dict(det_state_dict(shapes_of(m), seed))
        return cfg.GaussianDiffusion(m, image_size=16, timesteps=1000, sampling_timesteps=4, objective='pred_x0')
    online, ema = build(11).state_dict(), build(12).state_dict()
    assert set(online) == set(meta['diffusion_cfg_dim64_keys'])       # same key set as the reference's dim-64 model
    torch.save(_ema_shaped_checkpoint(online, ema, with_online), str(tmp_path / 'model-3.pt'))
    d = build(0)
    tr = ddpm.Trainer(d, 'DGM_Conditions', train_batch_size=2, results_folder=str(tmp_path))
    tr.load(3)
    assert tr.step == 1234 and int(tr.ema.step) == 1234 and bool(tr.ema.initted)
    for k, v in d.state_dict().items():
        assert torch.equal(v, online[k]), k                            # online weights: data['model'], untouched by the EMA
    assert tr.ema.ema_model is not tr.ema.online_model
    for k, v in tr.ema.ema_model.state_dict().items():
        assert torch.equal(v, ema[k]), k
    sd = tr.ema.state_dict()                                           # and it saves back with both prefixes
    assert torch.equal(sd['ema_model.model.init_conv.weight'], ema['model.init_conv.weight'])
    assert torch.equal(sd['online_model.model.init_conv.weight'], online['model.init_conv.weight'])
    # an untrained checkpoint (EMA == online) keeps sharing one copy: a sampling-only run pays for one set of weights
    torch.save(_ema_shaped_checkpoint(online, online, with_online), str(tmp_path / 'model-4.pt'))
    d2 = build(0)
    tr2 = ddpm.Trainer(d2, 'DGM_Conditions', train_batch_size=2, results_folder=str(tmp_path))
    tr2.load(4)
    assert tr2.ema.ema_model is tr2.ema.online_model


def test_trainer_split_batches_semantics():
    """accelerate's split_batches (DDP:1721-1722): True = train_batch_size is the global batch"""
    from dmhomo_amd import cfg, ddpm
    m = cfg.Unet(dim=8, channels=6, num_classes=1)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=10)
    tr = ddpm.Trainer(d, 'DGM_Conditions', train_batch_size=6)
    assert tr.batch_size == 6 and tr.rank_batch_size == 6 and tr.split_batches      # one process: the whole batch
    tr = ddpm.Trainer(d, 'DGM_Conditions', train_batch_size=6, split_batches=False)
    assert tr.rank_batch_size == 6


def test_bench_roofline_is_a_fraction_of_the_pipe_that_runs():
    """bench.py's `roofline` object from a synthetic HIP-event log: frac = achieved / peak with the peak of the pipe the
    kernel executes on (2.5 PFLOP/s fp16 MFMA / 3 executed FLOPs per algorithmic one), never above 1 for a physically
    possible time; the canonical launch's HBM-equivalent fraction; the sub-pixel convs kept out of frac"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class Ev:
        def __init__(self, ms=None):
            self.ms = ms

        def elapsed_time(self, other):
            return other.ms

    class A:
        bs, image_size = 25, 128
    # (e0, e1, k, stride, B, ho, wo, cin, cout, ups, prologue): two canonical 3x3 launches (with the GroupNorm + SiLU
    # prologue) at 190 us, the same shape without the prologue at 150 us, one 1x1, one sub-pixel conv
    log = [(Ev(), Ev(0.190), 3, 1, 50, 128, 128, 64, 64, 0, True), (Ev(), Ev(0.190), 3, 1, 50, 128, 128, 64, 64, 0, True),
           (Ev(), Ev(0.150), 3, 1, 50, 128, 128, 64, 64, 0, False),
           (Ev(), Ev(0.100), 1, 1, 50, 128, 128, 128, 64, 0, False), (Ev(), Ev(0.165), 3, 1, 50, 128, 128, 128, 64, 2, False)]
    r = bench.roofline(log, A)
    flop = 2.0 * 9 * 64 * 64 * 128 * 128 * 50
    assert r['launches'] == 3 and abs(r['achieved'] - 3 * flop / 0.530e-3 / 1e12) < 1e-6
    assert r['same_shape_without_prologue']['launches'] == 1
    assert abs(r['same_shape_without_prologue']['avg_launch_us'] - 150.0) < 1e-6
    empty = bench.roofline([(Ev(), Ev(0.1), 1, 1, 50, 128, 128, 128, 64, 0, False)], A)   # no 3x3 launch: no division by 0
    assert empty['frac'] is None and empty['achieved'] is None
    assert abs(r['peak'] - 2500.0 / 3) < 1e-9 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0 < r['frac'] < 1
    assert abs(r['executed']['frac'] - r['frac']) < 1e-12
    assert abs(r['hbm_frac_canonical'] - 419604224.0 / 190e-6 / 8e12) < 1e-9
    assert r['canonical_64to64_128sq']['launches'] == 2 and r['subpixel_upsample_convs']['TFLOP/s_algorithmic'] > 0
    assert r['traffic'] is None or r['traffic_source'].startswith('profiles/')


def test_bench_measurement_children_are_fresh_bounded_and_outside_the_rendezvous():
    """bench.py's measurement children (the rocprofv3 --pmc traffic passes, the exact-fp32 variant): their environment carries
    none of a torchrun parent's rendezvous variables (a one-rank child must not join the parent's live process group), and a
    child that hangs WITH a grandchild holding its pipes — rocprofv3 -> python is that shape — is killed as a process group:
    run_child raises TimeoutExpired promptly instead of blocking in the read after the kill."""
    import importlib.util
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    saved = dict(os.environ)
    try:
        os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29511',
                          TORCHELASTIC_RUN_ID='x', GROUP_RANK='0')
        env = bench.child_env(DMH_CONV3_VARIANT='6')
    finally:
        os.environ.clear()
        os.environ.update(saved)
    assert env['DMH_CONV3_VARIANT'] == '6'
    assert not any(k in env for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'GROUP_RANK'))
    assert not any(k.startswith('TORCHELASTIC_') for k in env)
    rc, out, err = bench.run_child([sys.executable, '-c', 'print("ok")'], 30)
    assert rc == 0 and out.strip() == 'ok'
    # parent sleeps, its child (same session, inherits the pipes) sleeps longer: both must be gone after the timeout
    code = ('import subprocess, sys, time\n'
            'p = subprocess.Popen([sys.executable, "-c", "import time; print(\\"grandchild\\", flush=True); time.sleep(600)"])\n'
            'print(p.pid, flush=True)\n'
            'time.sleep(600)\n')
    t0 = time.time()
    try:
        bench.run_child([sys.executable, '-c', code], 3)
        raise AssertionError('no timeout')
    except subprocess.TimeoutExpired:
        pass
    assert time.time() - t0 < 30
    # smi_sample never raises (no rocm-smi / no GPU here: an 'error' entry) — and starts NO program under a profiler preload
    # (rocm-smi is an env-python3 shebang script: under rocprofv3 that is an exec after GPU initialisation, which the pool forbids)
    assert isinstance(bench.smi_sample(0), dict)
    try:
        os.environ['ROCPROF_OUTPUT_PATH'] = '/tmp/x'
        assert bench.under_profiler() and 'skipped' in bench.smi_sample(0)
    finally:
        os.environ.pop('ROCPROF_OUTPUT_PATH')
    assert not bench.under_profiler()


def test_training_refuses_learned_sinusoidal_models():
    """train.UnetTrain derives its sinusoidal embedding width from time_mlp.1 — sized for learned_dim + 1 inputs on a Unet built
    with learned_sinusoidal_cond / random_fourier_features (CFG:175-190, 336-344): a silent K mismatch, unreachable through
    GaussianDiffusion (which refuses such a model, CFG:514-515) and refused loudly for anyone who gets there another way."""
    import pytest
    from dmhomo_amd import cfg, train
    for kw in (dict(learned_sinusoidal_cond=True), dict(random_fourier_features=True)):
        m = cfg.Unet(dim=8, dim_mults=(1, 2), channels=6, num_classes=1, **kw)
        with pytest.raises(NotImplementedError, match='learned_sinusoidal_cond'):
            train.UnetTrain(m)

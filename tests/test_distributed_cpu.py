"""CPU, world_size 2 over gloo: the N>1 plumbing of the sample-sharded path (weights broadcast as
scatter + all-gather, contiguous shards, rank-ordered gather, GPU-count-invariant noise)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from dmhomo_amd import distributed as D
    from dmhomo_amd import cfg
    r, w, device = D.init_from_env('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                      # different initial weights per rank
    m = cfg.Unet(dim=8, dim_mults=(1, 2), channels=6, num_classes=1)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=50, sampling_timesteps=4, objective='pred_x0')
    if rank == 1:
        d.betas.add_(1.0)                              # buffers travel too
    D.broadcast_module_(d, src=0)
    digest = float(sum(p.double().sum() for p in d.state_dict().values()))
    lo, hi = D.shard_bounds(5, rank, world)
    total = 7                                          # uneven shards: rank 0 owns 4 samples, rank 1 owns 3
    slo, shi = D.shard_bounds(total, rank, world)
    imgs = torch.arange(slo, shi, dtype=torch.uint8).reshape(-1, 1, 1, 1).expand(-1, 6, 2, 2).contiguous()
    homos = torch.arange(slo, shi, dtype=torch.float64).reshape(-1, 1, 1).expand(-1, 3, 3).contiguous()
    gi, gh = D.gather_records(imgs, homos, dst=0)
    # the key each rank derives for the sample-indexed noise generator (dmh_rng_indexed needs the GPU; its numpy
    # restatement stands in for the values here): same seed, this rank's slice of the global sample ids
    from oracle import rng as ORNG
    seed, ids = D.noise_key(7, total, rank, world, first_id=1000)
    assert seed == 7 and list(ids) == list(range(1000 + slo, 1000 + shi))
    n1 = torch.from_numpy(ORNG.randn(seed, ids, 5, (2, 3)))
    u1 = torch.from_numpy(ORNG.uniform(seed, ids, 6)[:, 0])
    # training: gradient averaging (the scale step is a HIP kernel on the GPU path; a torch multiply stands in here)
    grads = {'b.weight': torch.full((2, 3), float(rank + 1)), 'a.bias': torch.arange(4.) * (rank + 1)}
    avg = D.average_gradients(grads, lambda flat, sc: flat * sc)
    assert torch.equal(avg['b.weight'], torch.full((2, 3), 1.5)) and torch.equal(avg['a.bias'], torch.arange(4.) * 1.5)
    # accelerate's split_batches=True (the reference's default, DDP:1721-1722): train_batch_size is the GLOBAL batch
    from dmhomo_amd import ddpm
    assert ddpm.Trainer(d, 'DGM_Conditions', train_batch_size=6).rank_batch_size == 3
    assert ddpm.Trainer(d, 'DGM_Conditions', train_batch_size=6, split_batches=False).rank_batch_size == 6
    try:
        ddpm.Trainer(d, 'DGM_Conditions', train_batch_size=5)
        raise AssertionError('an indivisible global batch must be refused')
    except ValueError:
        pass
    q.put((rank, digest, (lo, hi), None if gi is None else gi[:, 0, 0, 0].tolist(),
           None if gh is None else gh[:, 0, 0].tolist(), n1.tolist(), u1.tolist()))   # plain lists: no shm fds
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, d0, b0, gi0, gh0, n0, u0), (r1, d1, b1, gi1, gh1, n1, u1) = res
    assert d0 == d1                                     # identical weights + buffers after the broadcast
    assert b0 == (0, 3) and b1 == (3, 5)                # contiguous, first ranks take the remainder
    assert gi0 == [0, 1, 2, 3, 4, 5, 6] and gh0 == [0., 1., 2., 3., 4., 5., 6.] and gi1 is None
    from dmhomo_amd import distributed as D
    # N-rank noise == the single-process noise, row for row: the ranks' keys concatenate to the one-process key
    from oracle import rng as ORNG
    seed, ids = D.noise_key(7, 7, 0, 1, first_id=1000)
    assert list(ids) == list(D.noise_key(7, 7, 0, 2, first_id=1000)[1]) + list(D.noise_key(7, 7, 1, 2, first_id=1000)[1])
    assert torch.equal(torch.tensor(n0 + n1), torch.from_numpy(ORNG.randn(seed, ids, 5, (2, 3))))
    assert torch.equal(torch.tensor(u0 + u1), torch.from_numpy(ORNG.uniform(seed, ids, 6)[:, 0]))


def _worker8(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from dmhomo_amd import distributed as D
    from dmhomo_amd import cfg
    r, w, device = D.init_from_env('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)
    m = cfg.Unet(dim=8, dim_mults=(1, 2), channels=6, num_classes=1)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=50, sampling_timesteps=4, objective='pred_x0')
    if rank == world - 1:
        d.betas.add_(1.0)
    D.broadcast_module_(d, src=0)          # the payload is cut into `world` pieces: scatter + all_gather at world 8
    digest = float(sum(p.double().sum() for p in d.state_dict().values()))
    total = 25                             # BASELINE configs[1]'s batch dealt to configs[2]'s 8 ranks: shards 4, 3, 3, ...
    lo, hi = D.shard_bounds(total, rank, world)
    imgs = torch.arange(lo, hi, dtype=torch.uint8).reshape(-1, 1, 1, 1).expand(-1, 6, 2, 2).contiguous()
    homos = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1, 1).expand(-1, 3, 3).contiguous()
    gi, gh = D.gather_records(imgs, homos, dst=0)
    seed, ids = D.noise_key(7, total, rank, world, first_id=500)
    q.put((rank, digest, (lo, hi), None if gi is None else gi[:, 0, 0, 0].tolist(),
           None if gh is None else gh[:, 0, 0].tolist(), seed, list(ids)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_world8_gloo():
    """world_size 8 — the process count of BASELINE configs[2] / configs[3] — on the host: weight payload in 8 pieces, uneven
    contiguous shards of a 25-sample job, the size-exchanged gather in rank order, the ranks' noise keys concatenating to the
    one-process key"""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert len({t[1] for t in res}) == 1                 # identical weights + buffers on all 8 ranks
    spans = [t[2] for t in res]
    assert spans == [(0, 4), (4, 7), (7, 10), (10, 13), (13, 16), (16, 19), (19, 22), (22, 25)]
    assert res[0][3] == list(range(25)) and res[0][4] == [float(i) for i in range(25)]
    assert all(t[3] is None and t[4] is None for t in res[1:])
    from dmhomo_amd import distributed as D
    assert all(t[5] == 7 for t in res)
    assert sum((t[6] for t in res), []) == list(D.noise_key(7, 25, 0, 1, first_id=500)[1])


def _ckpt_worker(rank, world, port, q, folder):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from dmhomo_amd import distributed as D
    from dmhomo_amd import cfg, ddpm
    D.init_from_env('gloo')
    torch.manual_seed(300 + rank)                      # every rank starts from different weights
    m = cfg.Unet(dim=8, dim_mults=(1, 2), channels=6, num_classes=1)
    d = cfg.GaussianDiffusion(m, image_size=16, timesteps=50, sampling_timesteps=4, objective='pred_x0')
    # only rank 0 is given the folder that holds the checkpoint: another rank touching the file would raise
    tr = ddpm.Trainer(d, 'DGM_Conditions', train_batch_size=2, results_folder=folder if rank == 0 else folder + '/absent')
    loaded = D.load_on_rank0_and_broadcast(tr, 7)
    dig = lambda mod: float(sum(p.double().sum() for p in mod.state_dict().values()))
    # a checkpoint that rank 0 cannot read must fail on EVERY rank (not leave the peers in a collective)
    try:
        D.load_on_rank0_and_broadcast(tr, 'absent')
        failed = False
    except (RuntimeError, FileNotFoundError):
        failed = True
    q.put((rank, loaded, tr.ema.ema_model is not tr.ema.online_model, dig(tr.model), dig(tr.ema.ema_model),
           int(tr.step), int(tr.ema.step), bool(tr.ema.initted), failed))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_checkpoint_is_read_by_rank0_only_and_broadcast(tmp_path):
    """scripts/dgm_sample.py's start-up (north_star: 'RCCL-over-xGMI broadcast of UNet weights'): rank 0 alone calls
    Trainer.load (DDP:1804-1826); the online AND the EMA copy (which differ in a trained checkpoint, and which
    Trainer.sample reads, DDP:1960) reach rank 1 through broadcast_module_, EMA copy in storage of its own on both."""
    from dmhomo_amd import cfg
    def build(seed):
        torch.manual_seed(seed)
        m = cfg.Unet(dim=8, dim_mults=(1, 2), channels=6, num_classes=1)
        return cfg.GaussianDiffusion(m, image_size=16, timesteps=50, sampling_timesteps=4, objective='pred_x0')
    online, ema = build(1).state_dict(), build(2).state_dict()
    e = {'initted': torch.tensor(True), 'step': torch.tensor(5)}
    e.update({'online_model.' + k: v.clone() for k, v in online.items()})
    e.update({'ema_model.' + k: v.clone() for k, v in ema.items()})
    torch.save({'step': 5, 'model': online, 'opt': None, 'ema': e, 'scaler': None, 'version': '1.0.0'},
               str(tmp_path / 'model-7.pt'))
    want_on = float(sum(p.double().sum() for p in online.values()))
    want_ema = float(sum(p.double().sum() for p in ema.values()))
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ckpt_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, loaded, own, d_on, d_ema, step, ema_step, initted, failed in res:
        assert loaded and own, (rank, loaded, own)
        assert d_on == want_on and d_ema == want_ema, (rank, d_on, want_on, d_ema, want_ema)
        # the counters a resumed Trainer.train loops on travel too, so the ranks agree on the number of steps left
        assert (step, ema_step, initted) == (5, 5, True), (rank, step, ema_step, initted)
        assert failed, rank


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment: the parent starts the ranks itself (README:14 /
    dgm_sample.py:13-18 start N processes by hand), relays rank 0's single JSON line and returns the children's exit
    code.  Run here in the script's test-only CPU plumbing mode (gloo; the sampling step has no CPU path)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['DMH_BENCH_PLUMBING_TEST'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--device', 'cpu', '--bs', '3',
                        '--steps', '2', '--warmup', '0'], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['rccl_ranks'] == 2 and res['records_in_rank_order'] and res['global_batch'] == 6
    # ... and the 8 ranks of BASELINE configs[2]
    env8 = dict(env, OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--device', 'cpu', '--bs', '3',
                        '--steps', '2', '--warmup', '0'], capture_output=True, text=True, env=env8, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert res['n_gpus'] == 8 and res['rccl_ranks'] == 8 and res['records_in_rank_order'] and res['global_batch'] == 24
    # the per-rank / per-phase diagnosis an N-rank line carries (bench.assemble_phases): one entry per rank, in rank order
    ph = res['phases']
    assert len(ph['per_rank_ms']) == 8 == len(ph['gather_ms']) == len(ph['local_ms']) == len(ph['smi'])
    assert ph['per_rank_ms_min'] <= ph['per_rank_ms_mean'] <= ph['per_rank_ms_max'] and 0 <= ph['straggler_rank'] < 8
    assert ph['per_rank_ms'][ph['straggler_rank']] == ph['per_rank_ms_max'] and ph['broadcast_ms'] > 0 and ph['broadcast_first_ms'] > 0
    assert [e['rank'] for e in ph['smi']] == list(range(8)) and all('before' in e and 'after' in e for e in ph['smi'])
    # outside the test the CPU mode refuses to run, and a failing child makes the launcher exit non-zero
    env.pop('DMH_BENCH_PLUMBING_TEST')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--device', 'cpu'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and 'test-only plumbing mode' in r.stderr


def test_shard_bounds_cover():
    from dmhomo_amd.distributed import shard_bounds
    for total in (0, 1, 7, 25, 200):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1

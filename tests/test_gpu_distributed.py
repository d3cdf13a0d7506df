"""the N > 1 data path of bench.py / scripts (dmhomo_amd/distributed.py) on the RCCL backend itself.  The pool's boxes have
one GPU, so the process group has ONE rank — enough to run every collective the path issues (scatter + all_gather weight
payload, broadcast of the integer buffers, all_gather of the shard sizes, gather of uint8 images and f64 homographies,
all_reduce of gradients and of the timing) through RCCL on device tensors: dtype / op support and stream semantics, which
the world-size-2 gloo tests on CPU cannot show.  Runs in a child process (a process group is global state)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ['DMH_ROOT'])
from dmhomo_amd import cfg, ops
from dmhomo_amd import distributed as D
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%s' % os.environ['DMH_PORT'], rank=0, world_size=1)
assert dist.get_backend() == 'nccl'
D.world_size = lambda: 2                      # take the collective branch of every helper (the group itself has 1 rank)
dev = torch.device('cuda', 0)
torch.manual_seed(3)
m = cfg.Unet(dim=8, dim_mults=(1, 2), channels=6, num_classes=1)
d = cfg.GaussianDiffusion(m, image_size=16, timesteps=50, sampling_timesteps=4, objective='pred_x0').to(dev)
before = {k: v.clone() for k, v in d.state_dict().items()}
D.broadcast_module_(d, src=0)                 # scatter + all_gather + broadcast over RCCL
for k, v in d.state_dict().items():
    assert torch.equal(v, before[k]), k
imgs = torch.arange(5 * 6 * 4 * 4, device=dev).reshape(5, 6, 4, 4).to(torch.uint8)
homos = torch.randn(5, 3, 3, device=dev, dtype=torch.float64)
gi, gh = D.gather_records(imgs, homos, dst=0)  # all_gather(int64) + gather(uint8) + gather(f64)
assert torch.equal(gi, imgs) and torch.equal(gh, homos)
grads = {'a': torch.randn(1000, device=dev), 'b': torch.randn(7, 3, device=dev)}
real_ws = dist.get_world_size
dist.get_world_size = lambda *a, **k: 2        # average_gradients' own guard
try:
    avg = D.average_gradients(grads, lambda flat, s: flat * s)
finally:
    dist.get_world_size = real_ws
for k in grads:
    assert torch.allclose(avg[k], grads[k] * 0.5), k   # one rank's sum, scaled by 1 / "2"
t = torch.tensor([1.5, 2.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)       # the timing reduction of bench.py
assert t.tolist() == [1.5, 2.5]
# the per-rank / per-phase diagnosis of an N-rank bench line travels as one all_gather_object (bench.assemble_phases): the
# object path of torch.distributed over RCCL (pickle -> byte tensors on the current device -> all_gather) on this stack
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.environ['DMH_ROOT'], 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
mine = {'rank': 0, 'ms_per_step': 280.5, 'local_ms': 279.0, 'gather_ms': 0.4, 'smi_before': bench.smi_sample(0),
        'smi_after': {'sclk clock speed': 2000.0}, 'host': 'h', 'device': 'cuda:0'}
got = [None]
dist.all_gather_object(got, mine)
assert got[0] == mine and isinstance(mine['smi_before'], dict) and 'error' not in mine['smi_before'], mine['smi_before']
ph = bench.assemble_phases(mine, 0, 1, 3.0, 30.0)
assert ph['per_rank_ms'] == [280.5] and ph['straggler_rank'] == 0 and ph['smi'][0]['before'] == mine['smi_before']
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print('RCCL-OK')
'''


def test_rccl_collectives_of_the_data_path_single_rank():
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, DMH_ROOT=root, DMH_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-c', CHILD], capture_output=True, text=True, env=env, cwd=root, timeout=600)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and 'RCCL-OK' in r.stdout


TWO_RANKS = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ['DMH_ROOT'])
from dmhomo_amd import cfg, ddpm, ops
from dmhomo_amd import distributed as D
rank, world, device = D.init_from_env()              # DMH_DIST_BACKEND=gloo, DMH_SHARE_GPU=1: both ranks on cuda:0
assert world == int(os.environ['DMH_WANT_WORLD']) and device.type == 'cuda' and dist.get_backend() == 'gloo'
dim, S, total, steps = (64, 128, 6, 4) if world == 2 else (8, 16, 25, 4)     # 25 samples over 8 ranks: shards of 4 and 3
torch.manual_seed(100 + rank)                        # different weights per rank: rank 0's must win
model = cfg.Unet(dim=dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
model.cfg_mode = 'streams'
model.dedup_dropped_rows = world > 2                 # (the 8-rank job also skips the conditional pass's dropped rows)
d = cfg.GaussianDiffusion(model, image_size=S, timesteps=1000, sampling_timesteps=steps, objective='pred_x0').to(device)
D.broadcast_module_(d, src=0)
d.hip_graph = True

def run(lo, hi, ids):
    conds = ddpm.SyntheticConditions(S, hi - lo, seed=1000 + lo, device=device)     # row i of the job: seed 1000 + i
    data, classes = next(conds)
    d.rng.key_by_sample(7, ids, device)
    img, _, fl = d.sample(classes, data[:, -5:-2].contiguous(), data[:, -2:].contiguous(), data[:, -6:-5].contiguous())
    return ops.to_uint8(img), ops.dlt_homography(fl)

lo, hi = D.shard_bounds(total, rank, world)
ids = D.key_noise_by_sample(d, 7, total, device=device)
assert list(ids) == list(range(lo, hi))
u8, hm = run(lo, hi, ids)
gi, gh = D.gather_records(u8, hm, dst=0)
t = torch.tensor([float(rank + 1)], device=device, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == float(world)
sizes = [None] * world
dist.all_gather_object(sizes, hi - lo)
assert sum(sizes) == total and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
if rank == 0:
    assert gi.shape == (total, 6, S, S) and gh.shape == (total, 3, 3)
    wi, wh = run(0, total, range(total))             # the same job in ONE process
    assert torch.equal(gi, wi), int((gi != wi).sum())
    assert torch.equal(gh, wh)
    assert not torch.equal(gi[0], gi[3])
    print('RANKS-OK', sizes)
else:
    assert gi is None and gh is None
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize('ranks', [2, 8])
def test_ranks_shard_a_job_and_reproduce_the_single_process_records(tmp_path, ranks):
    """BASELINE configs[2]'s data path with more than one rank on real hardware, as far as a one-GPU box allows: two
    processes (torch.distributed.run) time-share the GPU and talk over gloo (RCCL refuses two ranks on one device) — weights
    of rank 0 reach rank 1 (scatter + all-gather payload), each rank samples its shard of a 6-sample job at the real geometry
    (dim 64, 128x128, 'streams', per-step HIP graph, noise keyed by global sample index), the uint8 records and homographies
    are gathered in rank order — and rank 0 checks them BITWISE against the same job run in one process.
    ranks = 8 (BASELINE configs[2]'s process count; a small model so that 8 contexts fit beside each other): a 25-sample job,
    i.e. UNEVEN shards (4, 3, 3, ...): shard bounds, the size-exchanged gather and the payload at the world size the driver's
    8-GPU run uses, with the conditional pass's dropped rows skipped."""
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    script = tmp_path / 'two_ranks.py'
    script.write_text(TWO_RANKS)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(DMH_ROOT=root, DMH_DIST_BACKEND='gloo', DMH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2',
               DMH_WANT_WORLD=str(ranks))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={ranks}', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), str(script)], capture_output=True, text=True, env=env, cwd=root,
                       timeout=900)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and 'RANKS-OK' in r.stdout


def test_dgm_sample_short_last_batch_captures_once_per_shape(tmp_path):
    """scripts/dgm_sample.py over a condition set whose size is not a multiple of --bs (two epochs of 8 items at bs 3: batches
    of 3, 3, 2, 3, 3, 2 — the reference's DataLoader keeps the short last batch, DDP:1746-1752): the captured denoise steps
    are kept per batch shape (cfg.GaussianDiffusion.graph_cache_size), so the job captures twice, not twice per epoch."""
    import numpy as np
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    S = 32
    gen = torch.Generator().manual_seed(3)
    batches = [(torch.rand((n, 12, S, S), generator=gen), torch.zeros(n, dtype=torch.long)) for n in (3, 3, 2, 3, 3, 2)]
    torch.save(batches, str(tmp_path / 'cond.pt'))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'dgm_sample.py'), '-c', 'absent', '--s_step', '3', '--bs', '3',
                        '--exp', 'run0', '--image_size', str(S), '--seed', '5', '--batches', '6', '--conditions',
                        str(tmp_path / 'cond.pt')], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'rank 0: graph captures 2' in r.stdout, r.stdout[-1500:]
    recs = []
    for part in range(3):
        recs.extend(np.load(str(tmp_path / 'traindata' / 'run0' / 'dataset' / f'idx_0_rank_0_part_{part}_dm_cahomo_0.006k.npy'),
                            allow_pickle=True))
    assert [r_['imgs'].shape[0] for r_ in recs] == [3, 3, 2, 3, 3, 2]


@pytest.mark.parametrize('ranks', [1, 2, 8])
def test_dgm_sample_script_end_to_end(tmp_path, ranks):
    """scripts/dgm_sample.py (the counterpart of DGM/dgm_sample.py: same flags, same record files) as a user starts it —
    one process, and 2 / 8 processes under torch.distributed.run (time-sharing the GPU over gloo on this one-GPU box): rank 0
    alone would read the checkpoint (none here: the seeded initialisation is broadcast), every rank writes its part file in the
    reference's list-of-dict format, the records are uint8 (bs, 6, S, S) + float64 (bs, 3, 3), a repeated run reproduces them
    bit for bit, and — noise AND synthetic conditions being keyed by the global sample index — the N ranks' records are, between
    them, bit for bit the records ONE process writes in N times as many batches (rank r's batch b = the single process's
    batch b * N + r)."""
    import socket
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, 'scripts', 'dgm_sample.py')
    S, bs = (128, 3) if ranks <= 2 else (32, 1)
    base = ['-c', 'absent', '--s_step', '3', '--bs', str(bs), '--exp', 'run0', '--image_size', str(S), '--seed', '5']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2', PYTHONPATH=root)
    tag = f'{2 * bs / 1000}k'

    def run(workdir, n, batches):
        os.makedirs(workdir)
        args = base + ['--batches', str(batches)]
        if n == 1:
            cmd = [sys.executable, script] + args
            e = env
        else:
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
                   '127.0.0.1', '--master-port', str(port), script] + args
            e = dict(env, DMH_DIST_BACKEND='gloo', DMH_SHARE_GPU='1')
        r = subprocess.run(cmd, capture_output=True, text=True, env=e, cwd=workdir, timeout=1200)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        out = {}
        for rk in range(n):
            recs = []
            for part in range(batches // 2):
                f = os.path.join(workdir, 'traindata', 'run0', 'dataset', f'idx_0_rank_{rk}_part_{part}_dm_cahomo_{tag}.npy')
                assert os.path.exists(f), os.listdir(os.path.join(workdir, 'traindata', 'run0', 'dataset'))
                got = np.load(f, allow_pickle=True)
                assert len(got) == 2
                recs.extend(got)
            for rec in recs:
                assert rec['imgs'].dtype == np.uint8 and rec['imgs'].shape == (bs, 6, S, S)
                # (saveTrainPair squeezes the homographies, DDP:1675: a one-sample batch gives (3, 3))
                assert rec['homos'].dtype == np.float64 and rec['homos'].shape == ((bs, 3, 3) if bs > 1 else (3, 3))
            out[rk] = recs
        return out

    def same(x, y):
        return np.array_equal(x['imgs'], y['imgs']) and np.array_equal(x['homos'], y['homos'])
    a = run(str(tmp_path / 'a'), ranks, 2)
    assert not np.array_equal(a[0][0]['imgs'], a[0][1]['imgs'])               # the second batch draws new noise
    if ranks == 1:
        b = run(str(tmp_path / 'b'), 1, 2)
        assert all(same(x, y) for x, y in zip(a[0], b[0]))
        return
    assert not np.array_equal(a[0][0]['imgs'], a[1][0]['imgs'])
    one = run(str(tmp_path / 'one'), 1, 2 * ranks)[0]                         # the same job in ONE process
    for rk in range(ranks):
        for b_ in range(2):
            assert same(a[rk][b_], one[b_ * ranks + rk]), (rk, b_)


@pytest.mark.parametrize('ranks,extra', [(2, ['--bs', '4']), (8, ['--bs', '3', '--dim', '8', '--image_size', '16'])])
def test_bench_ranks_on_one_gpu(ranks, extra):
    """`python bench.py --gpus 2` (and `--gpus 8` on a small model) end to end on real hardware — the launcher starts two ranks, the weight payload travels, each
    rank samples its shard (per-step graph, keyed noise), records are gathered every step, the time is the maximum over
    ranks, rank 0 prints ONE line — with the two ranks time-sharing the box's one GPU over gloo (the driver's 8-GPU run is the
    same code over RCCL)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(DMH_DIST_BACKEND='gloo', DMH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(ranks), '--s_step', '4', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline', '--no-roofline'] + extra, capture_output=True, text=True, env=env,
                       cwd=root, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == ranks and d['rccl_ranks'] == ranks and d['backend'] == 'gloo' and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == int(extra[1]) * ranks and d['value'] > 0 and d['steps'] == 2
    assert d['variants']['dedup_dropped_rows']['value'] > 0
    # the per-rank / per-phase diagnosis: one entry per rank, the weight payload, local step and gather
    # timed apart (outside the timed region), one rocm-smi sample per rank before / after the timed loop
    ph = d['phases']
    assert len(ph['per_rank_ms']) == ranks == len(ph['gather_ms']) == len(ph['local_ms']) == len(ph['smi'])
    assert ph['per_rank_ms_min'] <= ph['per_rank_ms_mean'] <= ph['per_rank_ms_max'] <= d['ms_per_step'] * 1.001
    assert 0 <= ph['straggler_rank'] < ranks and ph['broadcast_ms'] > 0 and min(ph['local_ms']) > 0
    assert all(isinstance(e['after'], dict) for e in ph['smi'])
    print('[phases]', json.dumps(ph)[:1500])
    assert 'exact_fp32' not in d['variants']               # (N = 1 only: a child process of a one-rank run)
    assert d['config']['measured_rel_error']['f16x3'] < 2e-6


def test_bench_single_gpu_line_carries_exact_fp32_and_measured_error():
    """the N = 1 line (a small model): `variants.exact_fp32` from a fresh child process under DMH_CONV3_VARIANT=6 (value,
    ms_per_step, slower than or equal to the fp16-piece headline is NOT asserted on a toy model), `config.measured_rel_error`
    with the measured error of both arithmetics on the canonical conv shape against fp64 (both fp32-class), and `phases`."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.pop('DMH_CONV3_VARIANT', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--s_step', '4', '--steps', '2', '--warmup', '1', '--bs', '3',
                        '--dim', '8', '--image_size', '32', '--no-cpu-baseline', '--no-roofline'], capture_output=True, text=True,
                       env=env, cwd=root, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    ex = d['variants']['exact_fp32']
    assert 'error' not in ex, ex
    assert ex['value'] > 0 and ex['ms_per_step'] > 0 and ex['steps'] == 3 and ex['speedup_vs_value'] > 0
    err = d['config']['measured_rel_error']
    print('[parity] canonical conv, measured rel. error vs fp64:', err)
    assert 0 < err['f16x3'] < 2e-6 and 0 < err['exact_fp32'] < 2e-6
    ph = d['phases']
    assert len(ph['per_rank_ms']) == 1 and ph['per_rank_ms'][0] <= d['ms_per_step'] * 1.001 and ph['local_ms'][0] > 0
    assert isinstance(ph['smi'][0]['after'], dict)


def test_train_bench_eight_ranks_on_one_gpu():
    """BASELINE configs[3]'s process count: `python bench.py --workload train --gpus 8` (a small model, 2 images per rank, the 8
    ranks time-sharing the one GPU over gloo) — weights of rank 0 on every rank, forward + backward on the HIP kernels, ONE
    flat gradient all-reduce per optimiser step, clip + Adam + re-pack — and afterwards every rank holds bit for bit the same
    parameters"""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(DMH_DIST_BACKEND='gloo', DMH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='2')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--workload', 'train', '--gpus', '8', '--dim', '8',
                        '--image_size', '16', '--bs', '2', '--steps', '2', '--warmup', '1'], capture_output=True, text=True,
                       env=env, cwd=root, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['backend'] == 'gloo' and d['ranks_agree'] is True
    assert d['value'] > 0 and d['loss'] == d['loss'] and abs(d['loss']) < 1e3          # finite

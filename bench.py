#!/usr/bin/env python3
"""Headline benchmark: sampled images/sec of the DGM denoising hot path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], README inference config at 128x128): CFG Unet dim=64,
dim_mults (1,2,4,8), channels=6, GaussianDiffusion(image_size=128, timesteps=1000,
sampling_timesteps=32, objective='pred_x0'), cond_scale=3 (2 UNet forwards per denoise step,
run as one 2B batch), bs=25 per GPU.  One "step" = one pass of the hot path over one batch:
sample() (32 denoise steps) + the uint8 / homography record of Trainer.sample + (N>1) the
gather to rank 0.  Inputs (conditions, weights) are resident in HBM before the timed region;
weights are seeded random init (the trained DGM.pt is not available offline; speed is weight
independent), noise comes from the sample-indexed device generator (dmh_rng_indexed: row i of an N-rank run is row i
of the 1-GPU run).  Samples shard across ranks with no
data-path collective -> "scaling": "weak" (25 samples per GPU).

N > 1 without a torchrun environment: this process only LAUNCHES (it never touches the GPU): it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's JSON line and exits
with the child's code — the counterpart of the reference's N hand-started processes (README:14,
DGM/dgm_sample.py:13-18).

Prints ONE JSON line on rank 0, including
  roofline     the dominant kernel (stride-1 3x3 conv, conv_f16x3_kernel<3,3,1,0,...>: implicit GEMM on the fp16
               matrix cores, every fp32 product carried by three fp16 MFMAs under block scaling).
               achieved = ALGORITHMIC FLOPs (2*9*Cin*Cout*H*W*B) of its launches / their HIP-event
               durations;  peak = the dense fp16 MFMA peak of MI355X (2.5 PFLOP/s, MI355X_MICROARCH.md) / 3 executed
               FLOPs per algorithmic FLOP = 833 algorithmic TFLOP/s: the ceiling of this arithmetic on the pipe it
               runs on;  frac = achieved / peak (= executed fp16-MFMA FLOP/s / 2.5 PFLOP/s).  Beside it
               `hbm_frac_canonical`: the north_star's yardstick, algorithmic bytes of the canonical fused
               conv3x3+GN+SiLU 64->64 @128^2 launch / its time / 8 TB/s (target >= 0.30), and `vs_fp32_mfma_peak`
               (what an exact-fp32 MFMA kernel could reach at most).  HIP events sit on the launch stream around
               every dmh_conv2d of ONE EXTRA untimed step in batched mode (exclusive per-launch durations: in the
               default 'streams' mode the two CFG passes overlap); the timed region itself carries no events.
               traffic = HBM bytes per launch of that kernel, MEASURED BY THIS RUN (round 5): two child runs of this script
               under `rocprofv3 --pmc` (FETCH_SIZE x2, WRITE_SIZE: separate passes) after the timed region; the committed
               figure of profiles/ stays beside it (`traffic_committed`, `traffic_vs_committed`); `--no-traffic`, N > 1 or a
               failing profiler fall back to the committed figure, tagged with its source file
  cpu_baseline the oracle (a port of the reference's CPU path, bit-equal to it on the build host)
               timed on this box's host cores on a bounded sample (bs=2, s_step=4, same network).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MFMA_TFLOPS = 157.3     # dense fp32 MFMA = fp32 vector peak (MI355X_MICROARCH.md)
PEAK_FP16_MFMA_TFLOPS = 2500.0    # dense fp16 / bf16 MFMA peak (same guide); the pipe conv_f16x3_kernel executes on
PEAK_HBM_TBS = 8.0                # HBM3E peak (same guide)
F16X3_TERMS = 3                   # executed fp16 MFMA FLOPs per algorithmic FLOP (h2*g1s + h1*g2 + h1*g1)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as a child process tree and relay
    rank 0's JSON line.  This parent makes no HIP / torch.cuda call (it does not even import torch) and never
    re-execs itself; it exits non-zero when any rank failed."""
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL needs on this pool
    env.setdefault('OMP_NUM_THREADS', '8')
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        if out.startswith('{') and '"metric"' in out:
            line = out.strip()
        else:
            sys.stdout.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0 or line is None:
        print(f'bench.py: the {n}-rank child run failed (exit code {rc}, JSON line {"present" if line else "missing"})',
              file=sys.stderr)
        sys.exit(rc if rc != 0 else 1)


def cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo (SMT siblings counted once); None when it cannot be read"""
    try:
        cores, phys, cid = set(), None, None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    phys = line.split(':', 1)[1].strip()
                elif line.startswith('core id'):
                    cid = line.split(':', 1)[1].strip()
                elif not line.strip():
                    if cid is not None:
                        cores.add((phys, cid))
                    phys = cid = None
        if cid is not None:
            cores.add((phys, cid))
        return len(cores) or None
    except OSError:
        return None


def _oracle_forward_setup(dim, image_size, conds=None):
    import torch
    from oracle import unet as OU
    from detweights import det_state_dict, shapes_of
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    sd = det_state_dict(shapes_of(m))
    B = 2
    g = torch.Generator().manual_seed(1)
    if conds is not None and conds[0].shape[0] >= B:
        rgb, mask = conds[0][:B].contiguous(), conds[2][:B].contiguous()
    else:
        rgb = torch.rand((B, 3, image_size, image_size), generator=g)
        mask = (torch.rand((B, 1, image_size, image_size), generator=g) > 0.5).float()
    x = torch.randn((B, 6, image_size, image_size), generator=g)
    tt = torch.full((B,), 500, dtype=torch.long)
    classes = torch.zeros(B, dtype=torch.long)

    def forward():
        with torch.no_grad():
            OU.cfg_unet_forward(sd, x, tt, classes, rgb, mask, None)
    return forward


def cpu_probe(args):
    """child-process mode (`bench.py --cpu-probe N`): median seconds of 3 oracle UNet forwards (bs=2) at N threads, after one
    warm-up — the all-cores point of `cpu_baseline`, run apart so that the parent can bound it with a timeout"""
    import torch
    torch.set_num_threads(args.cpu_probe)
    fwd = _oracle_forward_setup(args.dim, args.image_size)
    fwd()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fwd()
        ts.append(time.perf_counter() - t0)
    print(json.dumps({'threads': args.cpu_probe, 'median_s': sorted(ts)[1], 'all_s': ts}))


def cpu_baseline(dim, image_size, seconds=12.0, slice_bs=0, conds=None):
    """oracle (kind 'port') on the host cores: bs=2, s_step=4 passes of the same network, reported as
    images/s at s_step=32 (cost per denoise step is constant, so x 4/32).  conds: (rgb_flow01, flow, mask) CPU tensors —
    rows of the synthetic conditions the GPU leg ran on.
    Thread count: oneDNN over-threads this bs=2 workload on a many-core host (8 threads 0.24 s, 16: 0.13 s, 32: 0.26 s,
    128: 1.24 s, 256: 104 s per UNet forward on the round-1 / round-4 boxes), so a sweep picks it — on the MEDIAN of three
    forwards per count (round 4 picked on one: a 1 % difference moved `value` by 25 % between runs), near-ties (< 3 %) going
    to the larger count — and the real sample is then timed at the best TWO counts; both are reported, `value` is the
    faster.  The all-physical-cores point SURVEY 8d asks for is measured in a child process under a timeout."""
    import torch
    from oracle import diffusion as OD
    from detweights import det_state_dict, shapes_of
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    sd = det_state_dict(shapes_of(m))
    buf = OD.schedule_buffers(1000, 'cosine')
    B, S = 2, 4
    g = torch.Generator().manual_seed(1)

    def conditions(n):
        if conds is not None and conds[0].shape[0] >= n:
            return tuple(t[:n].contiguous() for t in conds)
        return (torch.rand((n, 3, image_size, image_size), generator=g),
                torch.zeros((n, 2, image_size, image_size)),
                (torch.rand((n, 1, image_size, image_size), generator=g) > 0.5).float())
    rgb, flow, mask = conditions(B)
    classes = torch.zeros(B, dtype=torch.long)
    ncpu = os.cpu_count() or 1
    nphys = physical_cores() or ncpu
    fwd = _oracle_forward_setup(dim, image_size, conds)
    tried = {}
    for nt in sorted({n for n in (8, 16, 32, 64) if n <= ncpu} or {ncpu}):
        torch.set_num_threads(nt)
        fwd()                                            # warm-up at this count (oneDNN primitive caches, thread pool)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fwd()
            ts.append(time.perf_counter() - t0)
        tried[nt] = round(sorted(ts)[1], 4)
        if tried[nt] > 1.5 * min(tried.values()):
            break                                            # (past the optimum: the wider settings only get slower)
    best = min(tried.values())
    ranked = sorted(tried, key=lambda n: (tried[n] > 1.03 * best, tried[n] if tried[n] > 1.03 * best else -n))
    picks = ranked[:2]

    def sample_rate(nt):
        torch.set_num_threads(nt)

        def one():
            with torch.no_grad():
                OD.cfg_sample(sd, buf, classes, rgb, flow, mask, image_size=image_size, channels=6,
                              sampling_timesteps=S, objective='pred_x0')
        one()                                               # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            one()
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds / len(picks) or n >= 4:
                break
        return el / n, n
    timed = {nt: sample_rate(nt) for nt in picks}
    cores = min(timed, key=lambda n: timed[n][0])
    per_pass, n = timed[cores]
    torch.set_num_threads(cores)
    # the all-physical-cores point, bounded: a child process, killed after 90 s
    allc = {'threads': nphys}
    if nphys in tried:
        allc['forward_s'] = tried[nphys]
    else:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-probe', str(nphys), '--dim', str(dim),
                                '--image_size', str(image_size)], capture_output=True, text=True, timeout=90)
            allc['forward_s'] = json.loads(r.stdout.strip().splitlines()[-1])['median_s'] if r.returncode == 0 else None
            if r.returncode != 0:
                allc['error'] = r.stderr[-300:]
        except subprocess.TimeoutExpired:
            allc['forward_s'] = None
            allc['note'] = 'more than 90 s for warm-up + 3 forwards: over-threaded (the child was killed)'
    res = {'value': B * (S / 32.0) / per_pass, 'unit': 'images/s (s_step=32 equivalent)', 'cores': cores,
           'kind': 'port', 'cpu_model': cpu_model(), 'host_cores': ncpu, 'physical_cores': nphys,
           'threads_tried': {str(k): v for k, v in tried.items()},
           'threads_note': 'MEDIAN seconds of 3 oracle UNet forwards (bs=2) per thread count; the best two counts (ties < 3 % '
                           'to the larger) then run the real sample: `at_threads`; `cores` / `value` = the faster of them',
           'at_threads': {str(k): {'value': B * (S / 32.0) / v[0], 's_per_pass': round(v[0], 3), 'passes': v[1]}
                          for k, v in timed.items()},
           'all_physical_cores': allc,
           'conditions': 'synthetic homography conditions of the GPU leg (SURVEY 8d)' if conds is not None else 'torch.rand',
           'sample': f'oracle cfg_sample bs={B} s_step={S} {image_size}x{image_size} dim={dim}, {n} passes, '
                     f'{per_pass:.2f} s/pass = {per_pass / S * 1000:.0f} ms per denoise step (bs={B}); '
                     f'images/s scaled by 4/32 to s_step=32'}
    if allc.get('forward_s'):
        # one denoise step = 2 forwards; the sample's other work is < 1 % of it
        allc['value_estimate'] = B / (2 * allc['forward_s']) / 32.0
        allc['estimate_note'] = 'images/s (s_step=32 equivalent) from the forward time alone: bs / (2 forwards x 32 steps)'
    if slice_bs:
        # SURVEY 8d's second CPU row: a slice of the headline configuration itself (bs = 25, s_step = 2), extrapolated x16
        Bs, Ss = slice_bs, 2
        rgb2, flow2, mask2 = conditions(Bs)
        t1 = time.perf_counter()
        with torch.no_grad():
            OD.cfg_sample(sd, buf, torch.zeros(Bs, dtype=torch.long), rgb2, flow2,
                          mask2, image_size=image_size, channels=6, sampling_timesteps=Ss, objective='pred_x0')
        el2 = time.perf_counter() - t1
        res['slice'] = {'value': Bs * (Ss / 32.0) / el2, 'unit': 'images/s (s_step=32 equivalent)', 'cores': cores,
                        'sample': f'oracle cfg_sample bs={Bs} s_step={Ss} (one pass, {el2:.1f} s), EXTRAPOLATED x16 to '
                                  f's_step=32: the cost per denoise step is constant'}
    return res


def plumbing_only(args):
    """tests/test_distributed_cpu.py only: the N > 1 plumbing of this script (launcher -> ranks -> process group ->
    weight payload -> shards -> gather -> max-over-ranks timing -> ONE JSON line) on gloo / CPU, with the sampling
    step left out — there is no CPU path for it.  Refuses to run outside that test."""
    if os.environ.get('DMH_BENCH_PLUMBING_TEST') != '1':
        raise RuntimeError('bench.py --device cpu is a test-only plumbing mode (tests/test_distributed_cpu.py); the '
                           'benchmark itself needs MI355X GPUs — there is no CPU path')
    import torch
    import torch.distributed as dist
    from dmhomo_amd import cfg
    from dmhomo_amd import distributed as D
    rank, world, device = D.init_from_env('gloo')
    assert world == args.gpus and device.type == 'cpu'
    torch.manual_seed(rank)                               # different weights per rank before the payload
    model = cfg.Unet(dim=8, dim_mults=(1, 2), channels=6, num_classes=1)
    diffusion = cfg.GaussianDiffusion(model, image_size=16, timesteps=50, sampling_timesteps=4, objective='pred_x0')
    D.broadcast_module_(diffusion, src=0)
    digest = torch.tensor([float(sum(p.double().sum() for p in diffusion.state_dict().values()))], dtype=torch.float64)
    lo, hi = D.shard_bounds(args.bs * world, rank, world)
    imgs = torch.arange(lo, hi, dtype=torch.uint8).reshape(-1, 1, 1, 1).expand(-1, 6, 4, 4).contiguous()
    homos = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1, 1).expand(-1, 3, 3).contiguous()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gi, gh = D.gather_records(imgs, homos, dst=0)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ranks = torch.ones(1)
    dist.all_reduce(ranks)
    dmin, dmax = digest.clone(), digest.clone()
    dist.all_reduce(dmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(dmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        ok = (gi[:, 0, 0, 0].tolist() == list(range(args.bs * world)) and gh[:, 0, 0].tolist() ==
              [float(i) for i in range(args.bs * world)] and float(dmin) == float(dmax))
        print(json.dumps({'metric': 'plumbing only (no sampling: CPU test mode)', 'value': None, 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'rccl_ranks': int(ranks.item()),
                          'backend': dist.get_backend(), 'plumbing_only': True, 'records_in_rank_order': bool(ok),
                          'global_batch': args.bs * world, 'elapsed_s': float(t.item())}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--bs', type=int, default=25, help='samples per GPU')
    ap.add_argument('--s_step', type=int, default=32)
    ap.add_argument('--image_size', type=int, default=128)
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cfg-mode', default='streams', choices=['batched', 'streams'])
    ap.add_argument('--stream-splits', type=int, default=1, help="row sub-batches per CFG pass in 'streams' mode")
    ap.add_argument('--no-graph', action='store_true', help="launch every kernel from Python instead of replaying the "
                    "HIP graph of the sampling loop (cfg.GaussianDiffusion.hip_graph)")
    ap.add_argument('--no-roofline', action='store_true', help='skip the extra untimed step that carries the HIP events')
    ap.add_argument('--no-traffic', action='store_true',
                    help='quote roofline.traffic from the committed PMC profile instead of measuring it (two child runs under '
                         'rocprofv3 --pmc, ~40 s, after the timed region; single-GPU runs only)')
    ap.add_argument('--no-variants', action='store_true',
                    help='skip the extra steps (after the timed region) that time cfg.Unet.dedup_dropped_rows and report it '
                         'under "variants" (the headline `value` never includes it)')
    ap.add_argument('--dedup', action='store_true',
                    help='development / profiling: run the TIMED loop itself with cfg.Unet.dedup_dropped_rows (the line is then '
                         'labelled as that variant and is not the headline)')
    ap.add_argument('--workload', default='sample', choices=['sample', 'train'],
                    help="'train': the optimiser step of BASELINE configs[3] (16 images per GPU, gradients averaged over "
                         "RCCL) instead of the headline sampling loop; same launch contract, see tools/train_bench.py")
    ap.add_argument('--cpu-probe', type=int, default=0, help='internal: the child-process mode of cpu_baseline (see cpu_probe)')
    ap.add_argument('--device', default='cuda', choices=['cuda', 'cpu'],
                    help="'cpu' is the test-only plumbing mode of tests/test_distributed_cpu.py (raises elsewhere)")
    args = ap.parse_args()
    if args.cpu_probe:
        return cpu_probe(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])       # before anything touches torch / the GPU
    if args.device == 'cpu':
        return plumbing_only(args)
    if args.workload == 'train':
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import train_bench
        ns = argparse.Namespace(bs=16 if args.bs == 25 else args.bs, size=args.image_size, dim=args.dim, steps=args.steps,
                                warmup=args.warmup, accum=1, torch=False)
        line = train_bench.run(ns)
        if line is not None:
            print(json.dumps(line))
        return

    import torch
    import torch.distributed as dist
    from dmhomo_amd import cfg, ddpm, ops
    from dmhomo_amd import distributed as D

    rank, world, device = D.init_from_env()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert device.type == 'cuda', 'bench.py needs MI355X GPUs (no CPU path)'

    # ---- model: seeded init on every rank, then rank 0's weights win (one scatter+all-gather payload)
    torch.manual_seed(0)
    model = cfg.Unet(dim=args.dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    model.cfg_mode = args.cfg_mode
    model.stream_splits = args.stream_splits
    model.dedup_dropped_rows = bool(args.dedup)
    diffusion = cfg.GaussianDiffusion(model, image_size=args.image_size, timesteps=1000,
                                      sampling_timesteps=args.s_step, loss_type='l1', objective='pred_x0').to(device)
    D.broadcast_module_(diffusion, src=0)
    # (ONE denoise step is captured and replayed s_step times: any depth, the 250-step stress configuration included)
    use_graph = not args.no_graph
    diffusion.hip_graph = use_graph
    # ---- synthetic conditions of this rank's shard, resident in HBM
    lo, hi = D.shard_bounds(args.bs * world, rank, world)
    # noise keyed by GLOBAL sample index (dmh_rng_indexed inside the captured denoise step): row i of the N-rank job is
    # row i of the 1-GPU job, whatever N is (the reference's N hand-started processes each draw their own, unseeded stream)
    D.key_noise_by_sample(diffusion, 99, args.bs * world, device=device)
    conds = ddpm.SyntheticConditions(args.image_size, hi - lo, seed=1000 + lo, device=device)
    data, classes = next(conds)
    rgb_flow, flow, mask = data[:, -5:-2].contiguous(), data[:, -2:].contiguous(), data[:, -6:-5].contiguous()

    def step():
        img, _, fl = diffusion.sample(classes, rgb_flow, flow, mask)          # cond_scale=3 (CFG:714)
        imgs_u8 = ops.to_uint8(img)
        homos = ops.dlt_homography(fl)
        return D.gather_records(imgs_u8, homos, dst=0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    # ---- roofline leg: HIP events on the launch stream around every dmh_conv2d of ONE EXTRA untimed step with the
    # cond + null rows in one launch sequence (exclusive per-launch durations; same kernels, same shapes)
    log = None
    if not args.no_roofline and rank == 0:
        diffusion.hip_graph = False                      # (per-launch HIP events need eager launches)
        model.cfg_mode = 'batched'
        step_local = lambda: diffusion.sample(classes, rgb_flow, flow, mask)   # no collective: rank 0 only
        step_local()
        torch.cuda.synchronize()
        ops.CONV_LOG = []
        step_local()
        torch.cuda.synchronize()
        log, ops.CONV_LOG = ops.CONV_LOG, None
        model.cfg_mode = args.cfg_mode
    # ---- variants: reported beside the headline, never as it: the opt-in de-duplication of the conditional pass's
    # dropped rows (cfg.Unet.dedup_dropped_rows: identical samples, B + kept rows per denoise step instead of 2B), in the
    # headline's own configuration (captured step, cfg_mode as above), timed after the timed region
    dedup_elapsed, dedup_rows, dedup_steps = 0.0, None, max(2, min(args.steps, 5))
    if not args.no_variants:
        diffusion.hip_graph = use_graph
        model.dedup_dropped_rows = True
        step()                                           # (captures the de-duplicating step)
        fence()
        base_draw = int(diffusion.rng.state[1].item())
        t1 = time.perf_counter()
        for _ in range(dedup_steps):
            step()
        fence()
        dedup_elapsed = (time.perf_counter() - t1) / dedup_steps
        model.dedup_dropped_rows = bool(args.dedup)
        # the UNet rows those steps computed: the keep masks are a pure function of (seed, sample id, draw index) — a
        # sample() call makes 2 * s_step draws, the mask of denoise step k is draw 1 + 2k of the call
        st_, kept = diffusion.rng.state.clone(), 0
        for c_ in range(dedup_steps):
            for k_ in range(args.s_step):
                st_[1] = base_draw + c_ * 2 * args.s_step + 1 + 2 * k_
                kept += int(ops.rng_keep_mask(diffusion.rng.sample_ids, st_, 1 - model.cond_drop_prob).sum().item())
        dedup_rows = (hi - lo) + kept / (dedup_steps * args.s_step)
    rccl_ranks = 1
    if world > 1:
        t = torch.tensor([elapsed, dedup_elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dedup_elapsed = float(t[0].item()), float(t[1].item())
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                             # every rank took part in an RCCL collective
        rccl_ranks = int(ones.item())
        assert rccl_ranks == dist.get_world_size()

    if rank == 0:
        images = args.bs * world * args.steps
        res = {
            'metric': f'sampled images/sec ({args.image_size}x{args.image_size}, s_step={args.s_step})' + (
                ' [--dedup: the dedup_dropped_rows VARIANT, not the headline]' if args.dedup else ''),
            'value': images / elapsed, 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'ms_per_denoise_step': elapsed / args.steps / args.s_step * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'rccl_ranks': rccl_ranks, 'backend': (dist.get_backend() if world > 1 else None),
            'config': {'workload': f'DGM CFG-Unet dim={args.dim} {args.image_size}x{args.image_size} '
                                   f'bs={args.bs}/GPU s_step={args.s_step} cond_scale=3 ' + (
                                       '(BASELINE configs[1])' if (args.dim, args.image_size, args.bs, args.s_step) ==
                                       (64, 128, 25, 32) else '(BASELINE configs[4], stress; NOT the headline)' if
                                       (args.dim, args.image_size, args.bs, args.s_step) == (128, 256, 8, 250) else
                                       '(README / DGM/dgm_sample.py:28-38 geometry image_size=256; NOT the headline)' if
                                       (args.dim, args.image_size, args.bs, args.s_step) == (64, 256, 25, 32) else
                                       '(non-BASELINE configuration)'),
                       'global_batch': args.bs * world, 'sharding': f'samples x{world}, no data-path collective',
                       'cfg_mode': args.cfg_mode, 'hip_graph': use_graph,
                       'weights': 'seeded random init',
                       'noise': 'Philox4x32-10 keyed by (seed, global sample id, draw, element): dmh_rng_indexed, rows invariant to N',
                       'arithmetic': 'fp32 tensors; 3x3 / 1x1 convolutions and the attention projections multiply block-scaled '
                                     'fp16 pieces of the fp32 operands on the matrix cores (3 MFMAs per product block, '
                                     'fp32 accumulate, error at the fp32-accumulation level: DESIGN.md 3.1); '
                                     'DMH_CONV3_VARIANT=6 selects the exact-fp32 kernels'},
        }
        if not args.no_variants:
            res['variants'] = {'dedup_dropped_rows': {
                'value': args.bs * world / dedup_elapsed, 'unit': 'images/s', 'ms_per_step': dedup_elapsed * 1e3,
                'speedup_vs_value': (args.bs * world / dedup_elapsed) / (images / elapsed), 'steps': dedup_steps,
                'hip_graph': use_graph, 'cfg_mode': args.cfg_mode,
                'unet_rows_per_denoise_step': dedup_rows, 'unet_rows_per_denoise_step_full': 2 * (hi - lo),
                'rows_note': 'rank 0: its null-pass rows + the mean number of conditional rows kept per denoise step',
                'note': 'NOT the headline (`value` computes all 2B rows): opt-in cfg.Unet.dedup_dropped_rows — rows of the '
                        'conditional pass whose class was dropped (p = 0.5, CFG:404,415-425) equal their null-pass rows and '
                        'are not computed; the keep mask stays on the device (row subsets of include/dmhomo_hip.h) inside the '
                        'captured step; bitwise identical samples (tests/test_gpu_dedup.py)'}}
        if log:
            args.measure_traffic = (world == 1 and not args.no_traffic and
                                    (args.dim, args.image_size, args.bs) == (64, 128, 25))   # (the committed figure's workload)
            res['roofline'] = roofline(log, args)
        if not args.no_cpu_baseline:         # rank 0's host cores, whatever N is (the other ranks wait at the barrier)
            big = args.dim * args.image_size > 64 * 128
            torch.cuda.synchronize()
            res['cpu_baseline'] = cpu_baseline(args.dim, args.image_size, seconds=4.0 if big else 12.0,
                                               slice_bs=0 if big else args.bs,
                                               conds=(rgb_flow.cpu(), flow.cpu(), mask.cpu()))
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure_traffic(timeout=240):
    """HBM bytes per stride-1 3x3 launch, measured NOW: two child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE, then
    WRITE_SIZE: separate passes, as MI355X_MICROARCH.md prescribes; --kernel-trace only), on the workload tools/profile_round5.sh
    uses for the committed figure (one eager batched step at s_step = 2: same launches, same shapes).  Units / corrections as
    tools/pmc_traffic.py: counters in KiB, FETCH_SIZE doubled on gfx950.  -> (bytes per launch, launches) or raises."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which('rocprofv3') is None:
        raise RuntimeError('rocprofv3 not on PATH')
    kernel = 'conv_f16x3_kernel<3, 3, 1, 0'
    out = {}
    env = dict(os.environ, TMPDIR='/tmp')
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='dmh_pmc_', dir='/tmp')
        try:
            cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '-o', 't', '--',
                   sys.executable, os.path.abspath(__file__), '--steps', '1', '--warmup', '1', '--s_step', '2', '--cfg-mode',
                   'batched', '--no-variants', '--no-cpu-baseline', '--no-roofline', '--no-graph']
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd='/tmp', env=env)
            files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
            if r.returncode != 0 or not files:
                raise RuntimeError(f'rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-200:]}')
            vals = [float(row['Counter_Value']) for row in csv.DictReader(open(files[0]))
                    if row['Counter_Name'] == counter and kernel in row['Kernel_Name']]
            if not vals:
                raise RuntimeError(f'no {counter} rows for {kernel}')
            out[counter] = (sum(vals) / len(vals), len(vals))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2 * out['FETCH_SIZE'][0] * 1024 + out['WRITE_SIZE'][0] * 1024, out['FETCH_SIZE'][1]


def roofline(log, args):
    """the `roofline` object of the JSON line from the HIP-event log of one batched step (see the module docstring)."""
    fl3 = ms3 = n3 = by3 = 0.0      # stride-1 3x3 launches (the dominant kernel): 3 executed fp16 FLOPs per algorithmic one
    flc = msc = nc = 0.0            # of those, the canonical 64->64 @ image_size^2 with the GN+SiLU prologue
    flu = msu = 0.0                 # sub-pixel Upsample convs (16 of every 36 taps executed): reported, not in `frac`
    flq = msq = nq = 0.0            # the same shape WITHOUT the prologue (block 1 of a ResnetBlock): reported beside it
    for e0, e1, k, stride, B, ho, wo, cin, cout, ups, pro in log:
        if k != 3:
            continue
        ms = e0.elapsed_time(e1)
        fl = 2.0 * 9 * cin * cout * ho * wo * B
        if ups == 2:
            flu, msu = flu + fl, msu + ms
            continue
        fl3, ms3, n3 = fl3 + fl, ms3 + ms, n3 + 1
        by3 += 4.0 * B * ho * wo * (cin + cout) + 4 * (9 * cin * cout + 3 * cout) + 8 * B * cout     # SURVEY 8d, per launch
        if (cin, cout, ho) == (64, 64, args.image_size):
            if pro:
                flc, msc, nc = flc + fl, msc + ms, nc + 1
            else:
                flq, msq, nq = flq + fl, msq + ms, nq + 1
    if not ms3:                      # (a configuration without a stride-1 3x3 launch in the log)
        return {'bound': 'mfma', 'achieved': None, 'peak': PEAK_FP16_MFMA_TFLOPS / F16X3_TERMS, 'unit': 'TFLOP/s',
                'frac': None, 'traffic': None, 'note': 'no stride-1 3x3 conv launch was logged'}
    ach = fl3 / (ms3 * 1e-3) / 1e12
    peak = PEAK_FP16_MFMA_TFLOPS / F16X3_TERMS
    rows = 2 * args.bs
    canon_bytes = 4.0 * rows * args.image_size ** 2 * (64 + 64) + 4 * (9 * 64 * 64 + 3 * 64) + 8 * rows * 64
    canon_us = msc / max(nc, 1) * 1e3
    traffic, traffic_src = None, None
    for name in ('r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        tpath = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(tpath):       # FETCH_SIZE / WRITE_SIZE passes of rocprofv3 (tools/profile_round.sh)
            with open(tpath) as f:
                traffic = json.load(f).get('hbm_bytes_per_launch')
            traffic_src = 'profiles/' + name
            break
    traffic_note = ('HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes, averaged over the 3x3 '
                    'launches of a batched step); from the committed profile, not measured by this run')
    committed = traffic
    if getattr(args, 'measure_traffic', False):
        # measured by THIS run (two child processes under rocprofv3 --pmc, after the timed region); the committed figure stays
        # beside it, with the deviation — a traffic regression now shows in the driver's own line
        try:
            t0 = time.perf_counter()
            traffic, nl = measure_traffic()
            traffic_src = 'measured by this run: 2 x rocprofv3 --pmc child passes of bench.py (batched, eager, s_step 2)'
            traffic_note = (f'HBM bytes per launch: FETCH_SIZE x2 + WRITE_SIZE (separate passes), averaged over {nl} stride-1 3x3 '
                            f'launches; {time.perf_counter() - t0:.0f} s for the two passes')
        except Exception as e:                       # (no profiler on the box, a refused counter, a timeout): quote the file
            traffic_note += f' [live measurement failed: {str(e)[:160]}]'
    return {
        'kernel': 'conv_f16x3_kernel<3,3,1,0,...> (stride-1 3x3 conv, implicit GEMM, 3 x v_mfma_f32_16x16x32_f16 per fp32 '
                  'product block, fp32 accumulate)',
        'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
        'peak_note': 'algorithmic TFLOP/s; peak = 2500 dense fp16 MFMA TFLOP/s / 3 executed FLOPs per algorithmic FLOP',
        'executed': {'pipe': 'fp16 MFMA', 'TFLOP/s': ach * F16X3_TERMS, 'peak': PEAK_FP16_MFMA_TFLOPS,
                     'frac': ach * F16X3_TERMS / PEAK_FP16_MFMA_TFLOPS},
        'vs_fp32_mfma_peak': ach / PEAK_FP32_MFMA_TFLOPS,
        'traffic': traffic, 'traffic_source': traffic_src, 'traffic_note': traffic_note,
        'traffic_committed': committed,
        'traffic_vs_committed': (traffic / committed) if (traffic and committed) else None,
        'algorithmic_bytes_per_launch_avg': by3 / max(n3, 1),
        'traffic_vs_algorithmic_note': 'the canonical launch alone moves 1.02x its algorithmic bytes (profiles/r05_pmc_canonical.json); '
                                       'the average over ALL stride-1 3x3 launches sits ~1.2x above the SURVEY 8d sum because the <= 32^2 '
                                       'levels fetch their (small) input once per 128-channel output tile — those tiles run on different '
                                       'XCDs, i.e. behind different L2s, so that each XCD keeps its weight slice resident (512->512@16^2: 130 MB '
                                       'fetched for 36 MB of algorithmic reads, at < 1 TB/s: those launches are latency-bound, DESIGN 3.1)',
        'measured_in': 'one extra untimed step, cfg_mode=batched, HIP events on the launch stream',
        'launches': int(n3), 'avg_launch_us': ms3 / max(n3, 1) * 1e3,
        'algorithmic_flop_per_launch': fl3 / max(n3, 1),
        'same_shape_without_prologue': {'launches': int(nq), 'avg_launch_us': (msq / nq * 1e3) if nq else None},
        'canonical_64to64_128sq': {'launches': int(nc), 'avg_launch_us': canon_us,
                                   'what': 'conv3x3 64->64 at image_size^2 WITH the fused GroupNorm+SiLU prologue (block 2 of '
                                           'a ResnetBlock) and the GroupNorm partials in the epilogue',
                                   'TFLOP/s': (flc / (msc * 1e-3) / 1e12) if msc else None,
                                   'algorithmic_bytes_per_launch': canon_bytes,
                                   'GB/s_algorithmic': (canon_bytes / (canon_us * 1e-6) / 1e9) if msc else None},
        'hbm_frac_canonical': (canon_bytes / (canon_us * 1e-6) / 1e12 / PEAK_HBM_TBS) if msc else None,
        'subpixel_upsample_convs': {'TFLOP/s_algorithmic': (flu / (msu * 1e-3) / 1e12) if msu else None,
                                    'note': 'Upsample+conv3x3 in sub-pixel form executes 16 of every 36 taps; kept out of frac'},
    }


if __name__ == '__main__':
    main()

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
independent), noise comes from the device Philox generator.  Samples shard across ranks with no
data-path collective -> "scaling": "weak" (25 samples per GPU).

Prints ONE JSON line on rank 0, including
  roofline     the dominant kernel (3x3 conv, conv_f16x3_kernel: implicit GEMM on the fp16 matrix cores, every
               fp32 product carried by three fp16 MFMAs under block scaling): ALGORITHMIC FLOPs
               (2*9*Cin*Cout*H*W*B) of its launches / their HIP-event durations, vs the
               157.3 TFLOP/s dense fp32 MFMA peak of MI355X (MI355X_MICROARCH.md; the dtype the path
               computes in is f32).  The kernel does not use the fp32 MFMA, so `frac` can exceed 1;
               `executed` is the same time priced on the pipe that actually runs: 3 fp16 MFMA FLOPs per
               algorithmic FLOP vs the 2.5 PFLOP/s dense fp16 peak.  With
               --cfg-mode streams (default) the two CFG passes run on two HIP streams and their kernels
               overlap, so per-launch durations are not exclusive: the roofline leg is then measured on
               one extra, untimed step in batched mode (same kernels, same shapes, 2B rows per launch)
  cpu_baseline the oracle (a port of the reference's CPU path, bit-equal to it on the build host)
               timed on this box's host cores on a bounded sample (bs=2, s_step=4, same network).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_FP16_MFMA_TFLOPS = 2500.0    # dense fp16 / bf16 MFMA peak (same guide); the executed pipe of conv_f16x3_kernel


def cpu_baseline(dim, image_size, seconds=12.0):
    """oracle (kind 'port') on the host cores: bs=2, s_step=4 passes of the same network, reported as
    images/s at s_step=32 (cost per denoise step is constant, so x 4/32)."""
    from oracle import diffusion as OD
    from detweights import det_state_dict, shapes_of
    from dmhomo_amd import cfg
    m = cfg.Unet(dim=dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    sd = det_state_dict(shapes_of(m))
    buf = OD.schedule_buffers(1000, 'cosine')
    B, S = 2, 4
    g = torch.Generator().manual_seed(1)
    rgb = torch.rand((B, 3, image_size, image_size), generator=g)
    mask = (torch.rand((B, 1, image_size, image_size), generator=g) > 0.5).float()
    flow = torch.zeros((B, 2, image_size, image_size))
    classes = torch.zeros(B, dtype=torch.long)
    # 16 threads is the fastest setting for this bs=2 workload on the 128-core GPU-box host (8: 0.24 s,
    # 16: 0.13 s, 32: 0.26 s, 128: 1.24 s per UNet forward; tools/cpu_threads.py) — oneDNN over-threads beyond that
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)

    def one():
        with torch.no_grad():
            OD.cfg_sample(sd, buf, classes, rgb, flow, mask, image_size=image_size, channels=6,
                          sampling_timesteps=S, objective='pred_x0')
    one()                                               # warm-up (oneDNN primitive caches)
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 6:
            break
    per_pass = el / n
    return {'value': B * (S / 32.0) / per_pass, 'unit': 'images/s (s_step=32 equivalent)', 'cores': cores,
            'kind': 'port',
            'sample': f'oracle cfg_sample bs={B} s_step={S} {image_size}x{image_size} dim={dim}, {n} passes, '
                      f'{per_pass:.2f} s/pass = {per_pass / S * 1000:.0f} ms per denoise step (bs={B}); '
                      f'images/s scaled by 4/32 to s_step=32'}


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
    ap.add_argument('--no-conv-events', action='store_true', help='skip the per-launch HIP events')
    ap.add_argument('--variants', action='store_true',
                    help='also time the opt-in dedup_dropped_rows mode (3 extra steps) and report it under "variants"')
    ap.add_argument('--workload', default='sample', choices=['sample', 'train'],
                    help="'train': the optimiser step of BASELINE configs[3] (16 images per GPU, gradients averaged over "
                         "RCCL) instead of the headline sampling loop; same launch contract, see tools/train_bench.py")
    args = ap.parse_args()
    if args.workload == 'train':
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import train_bench
        ns = argparse.Namespace(bs=16 if args.bs == 25 else args.bs, size=args.image_size, dim=args.dim, steps=args.steps,
                                warmup=args.warmup, accum=1, torch=False)
        line = train_bench.run(ns)
        if line is not None:
            print(json.dumps(line))
        return

    import torch.distributed as dist
    from dmhomo_amd import cfg, ddpm, ops
    from dmhomo_amd import distributed as D

    rank, world, device = D.init_from_env()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)'
    assert device.type == 'cuda', 'bench.py needs MI355X GPUs (no CPU path)'

    # ---- model: seeded init on every rank, then rank 0's weights win (one scatter+all-gather payload)
    torch.manual_seed(0)
    model = cfg.Unet(dim=args.dim, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1)
    model.cfg_mode = args.cfg_mode
    model.stream_splits = args.stream_splits
    diffusion = cfg.GaussianDiffusion(model, image_size=args.image_size, timesteps=1000,
                                      sampling_timesteps=args.s_step, loss_type='l1', objective='pred_x0').to(device)
    D.broadcast_module_(diffusion, src=0)
    torch.manual_seed(99 + rank)                         # device Philox stream for the noise

    # ---- synthetic conditions of this rank's shard, resident in HBM
    lo, hi = D.shard_bounds(args.bs * world, rank, world)
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
    ops.CONV_LOG = None if args.no_conv_events else []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    log, ops.CONV_LOG = ops.CONV_LOG, None
    roofline_mode = args.cfg_mode
    if log is not None and args.cfg_mode == 'streams':
        # exclusive per-launch durations: one extra untimed step with the cond+null rows in ONE launch sequence
        model.cfg_mode = 'batched'
        step()
        fence()
        ops.CONV_LOG = []
        step()
        fence()
        log, ops.CONV_LOG = ops.CONV_LOG, None
        model.cfg_mode = args.cfg_mode
        roofline_mode = 'batched (extra untimed step)'
    # ---- --variants: reported beside the headline, never as it: the opt-in de-duplication of the conditional pass's
    # dropped rows (cfg.Unet.dedup_dropped_rows: identical outputs, B + kept rows per denoise step instead of 2B)
    dedup_elapsed = 0.0
    if args.variants:
        model.cfg_mode, model.dedup_dropped_rows = 'batched', True
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        fence()
        dedup_elapsed = (time.perf_counter() - t1) / 2
        model.cfg_mode, model.dedup_dropped_rows = args.cfg_mode, False
    if world > 1:
        t = torch.tensor([elapsed, dedup_elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dedup_elapsed = float(t[0].item()), float(t[1].item())

    if rank == 0:
        images = args.bs * world * args.steps
        res = {
            'metric': f'sampled images/sec ({args.image_size}x{args.image_size}, s_step={args.s_step})', 'value': images / elapsed, 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'ms_per_denoise_step': elapsed / args.steps / args.s_step * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'DGM CFG-Unet dim={args.dim} {args.image_size}x{args.image_size} '
                                   f'bs={args.bs}/GPU s_step={args.s_step} cond_scale=3 ' + (
                                       '(BASELINE configs[1])' if (args.dim, args.image_size, args.bs, args.s_step) ==
                                       (64, 128, 25, 32) else '(BASELINE configs[4], stress)' if
                                       (args.dim, args.image_size, args.bs, args.s_step) == (128, 256, 8, 250) else
                                       '(non-BASELINE configuration)'),
                       'global_batch': args.bs * world, 'sharding': f'samples x{world}, no data-path collective',
                       'weights': 'seeded random init', 'noise': 'device Philox',
                       'arithmetic': 'fp32 tensors; 3x3 / 1x1 convolutions and the attention projections multiply block-scaled '
                                     'fp16 pieces of the fp32 operands on the matrix cores (3 MFMAs per product block, '
                                     'fp32 accumulate, error at the fp32-accumulation level: DESIGN.md 3.1); '
                                     'DMH_CONV3_VARIANT=6 selects the exact-fp32 kernels'},
        }
        if args.variants:
            res['variants'] = {'dedup_dropped_rows': {
                'value': args.bs * world / dedup_elapsed, 'unit': 'images/s', 'ms_per_step': dedup_elapsed * 1e3,
                'note': 'NOT the headline: opt-in cfg.Unet.dedup_dropped_rows — rows of the conditional pass whose class '
                        'was dropped (p = 0.5, CFG:404) equal their null-pass rows and are computed once; bitwise identical '
                        'samples (tests/test_gpu_unet.py::test_fullsize_rows_independent_and_cfg_modes_agree)'}}
        if log:
            fl3 = ms3 = n3 = 0.0
            flc = msc = nc = 0.0
            flx = 0.0                     # executed fp16-MFMA FLOPs: 3 per algorithmic FLOP; the sub-pixel Upsample convs
            for e0, e1, k, stride, B, ho, wo, cin, cout, ups in log:      # (upsample2 == 2) execute 16 of every 36 taps
                if k != 3:
                    continue
                ms = e0.elapsed_time(e1)
                fl = 2.0 * 9 * cin * cout * ho * wo * B
                fl3, ms3, n3 = fl3 + fl, ms3 + ms, n3 + 1
                flx += fl * 3.0 * (16.0 / 36.0 if ups == 2 else 1.0)
                if (cin, cout, ho) == (64, 64, args.image_size):
                    flc, msc, nc = flc + fl, msc + ms, nc + 1
            ach = fl3 / (ms3 * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
            if os.path.exists(tpath):       # FETCH_SIZE / WRITE_SIZE passes of rocprofv3 over this command
                with open(tpath) as f:
                    traffic = json.load(f)
            res['roofline'] = {
                'kernel': 'conv_f16x3_kernel<3,3,...> (3x3 conv, implicit GEMM, 3 x v_mfma_f32_16x16x32_f16 per fp32 '
                          'product block, fp32 accumulate)',
                'bound': 'mfma', 'achieved': ach, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ach / PEAK_FP32_MFMA_TFLOPS,
                'executed': {'pipe': 'fp16 MFMA (3 executed FLOPs per algorithmic FLOP; the sub-pixel Upsample convs run 16 of '
                                     'every 36 taps)', 'TFLOP/s': flx / (ms3 * 1e-3) / 1e12,
                             'peak': PEAK_FP16_MFMA_TFLOPS, 'frac': flx / (ms3 * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS},
                'traffic': traffic, 'measured_in': roofline_mode,
                'launches': int(n3), 'avg_launch_us': ms3 / n3 * 1e3,
                'canonical_64to64_128sq': {'launches': int(nc), 'avg_launch_us': msc / max(nc, 1) * 1e3,
                                           'TFLOP/s': (flc / (msc * 1e-3) / 1e12) if msc else None,
                                           'GB/s_algorithmic': (nc * (4.0 * 2 * args.bs * args.image_size ** 2 * 128
                                                                      + 4 * (9 * 64 * 64 + 3 * 64))
                                                                / (msc * 1e-3) / 1e9) if msc else None},
            }
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(args.dim, args.image_size)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

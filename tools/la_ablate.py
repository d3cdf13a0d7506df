"""diagnostic: what each piece of the two fused LinearAttention passes is worth, measured by switching it off
(needs `make -C dmhomo_amd/csrc stamps`: the LA_ABL bits of linattn_fused.hip exist in that build only).

    python tools/la_ablate.py [--rows 50] > profiles/r05_linattn_ablation.txt

Times are HIP-event averages over launches that rotate through > 768 MiB of distinct inputs (every launch streams its x from
HBM, as inside the sampling step).  The diagnostic build carries the cycle stamps too, so its ablate = 0 line is a few percent
slower than the product library; the DIFFERENCES are what the table is for.  An ablated launch computes garbage."""
import argparse
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmhomo_amd import _lib
PRODUCT = _lib.LIB_PATH
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops
from dmhomo_amd._lib import call, ptr, lib

ap = argparse.ArgumentParser()
ap.add_argument('--rows', type=int, default=50)
ap.add_argument('--reps', type=int, default=24)
args = ap.parse_args()
dev = torch.device('cuda', 0)
h = lib()
h.dmh_la_set_ablate.restype = ctypes.c_int
h.dmh_la_set_ablate.argtypes = [ctypes.c_int]

CASES = [(0, 'full'), (1, 'second fp16 piece of every split not formed'), (2, 'no exponentials'),
         (4, 'pass 2: no LayerNorm + residual + store'), (8, 'x / statistics from one cached line (no HBM stream)'),
         (16, 'no projection MFMAs'), (1 | 2, 'no second pieces, no exponentials'),
         (8 | 4, 'one cached line, no LN + store (pass 2: no HBM traffic at all)'),
         (1 | 2 | 16, 'no second pieces, no exponentials, no projection MFMAs'),
         (1 | 2 | 4 | 8 | 16, 'everything off: staging + barriers + remaining MFMAs')]


def run(C, H, B, fuse):
    n = H * H
    nbytes = B * n * C * 4
    nrot = min(64, max(3, -(-3 * 2 ** 28 // nbytes)))
    xs = [torch.randn((B, H, H, C), device=dev) for _ in range(nrot)]
    stats = []
    for x in xs:
        st = torch.empty((B, n, 2), device=dev)
        call('dmh_pixel_stats', ptr(x), ptr(st), B * n, C, 1e-5, None, 0)
        stats.append(st)
    g = torch.ones(C, device=dev)
    pla = ops.PackedLinAttn(torch.randn((384, C, 1, 1), device=dev) * C ** -0.5)
    plo = ops.PackedLinAttnOut(torch.randn((64, 128, 1, 1), device=dev) * 0.1, torch.zeros(64, device=dev), torch.ones(64, device=dev))
    ns = lib().dmh_linattn_fused_splits(B, n)
    partial = torch.empty((B, ns, 4, 1088), device=dev)
    ctx = torch.randn((B, 4, 32, 32), device=dev) * 0.01
    y = torch.empty((B, H, H, 64 if fuse else 128), device=dev)

    def p1(i):
        call('dmh_linattn_fused_context', ptr(xs[i]), ptr(stats[i]), ptr(g), ptr(pla.wpack), ptr(partial), B, n, C, None)

    def p2(i):
        if fuse:
            call('dmh_linattn_fused_apply_out', ptr(xs[i]), ptr(stats[i]), ptr(g), ptr(pla.wpack), ptr(ctx), ptr(plo.wpack),
                 ptr(plo.bias), ptr(plo.ln_g), ptr(y), B, n, C, 32 ** -0.5, 1e-5, None)
        else:
            call('dmh_linattn_fused_apply', ptr(xs[i]), ptr(stats[i]), ptr(g), ptr(pla.wpack), ptr(ctx), ptr(y), B, n, C,
                 32 ** -0.5, None)

    def timed(f):
        for i in range(3):
            f(i % nrot)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(args.reps):
            f(i % nrot)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / args.reps
    alg1 = 4.0 * B * n * C + 8.0 * B * n                      # pass 1 reads x and the statistics
    alg2 = 4.0 * B * n * (C + (C if fuse else 0) + (64 if fuse else 128)) + 8.0 * B * n
    print(f'C={C} {H}x{H} B={B}: pass 1 = linattn_kv{"_ring" if C == 64 else ""}_kernel (algorithmic {alg1 / 1e6:.0f} MB), '
          f'pass 2 = linattn_qo_kernel<{"true" if fuse else "false"}> (algorithmic {alg2 / 1e6:.0f} MB; '
          f'{"x twice (staging + residual), y once" if fuse else "x once, out once"})')
    # three round-robin sweeps over the cases (clock / power state drifts over a run: a case is compared with its neighbours in
    # time), median per case; a block of untimed launches first
    for i in range(40):
        p1(i % nrot)
        p2(i % nrot)
    res = {bits: ([], []) for bits, _ in CASES}
    for sweep in range(3):
        for bits, _ in CASES:
            assert h.dmh_la_set_ablate(bits) == 0
            res[bits][0].append(timed(p1))
            res[bits][1].append(timed(p2))
    med = {b: (sorted(v[0])[1], sorted(v[1])[1]) for b, v in res.items()}
    base = med[0]
    for bits, name in CASES:
        t1, t2 = med[bits]
        print(f'  ablate={bits:2d} {name:62s} pass 1 {t1:7.1f} us ({t1 - base[0]:+6.1f})   pass 2 {t2:7.1f} us ({t2 - base[1]:+6.1f})',
              flush=True)
    print(f'  (spread of the ablate = 0 line over the three sweeps: pass 1 {min(res[0][0]):.1f} .. {max(res[0][0]):.1f} us, '
          f'pass 2 {min(res[0][1]):.1f} .. {max(res[0][1]):.1f} us)')
    h.dmh_la_set_ablate(0)
    print(f'  HBM time of the algorithmic bytes at 8 TB/s: pass 1 {alg1 / 8e6:.1f} us, pass 2 {alg2 / 8e6:.1f} us')


run(64, 128, args.rows, True)
run(128, 64, args.rows, False)

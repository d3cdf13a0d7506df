"""diagnostic: conv_f16x3_kernel with phases ablated (needs `make -C dmhomo_amd/csrc stamps`): which part of the launch
time each phase accounts for.  Results are wrong by construction; only the times matter.
    python tools/f16_ablate.py
ablate bits (DMH_WINO_ABLATE, read per launch by the stamps build): 1 no split / LDS write, 2 no matrix phase,
4 no epilogue, 8 weight fragments loaded once (no B stream), 16 A fragments read once per chunk (no LDS reads),
32 the halo prefetch of chunks >= 1 reads one cached line, 64 no halo prefetch of chunks >= 1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmhomo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops

dev = torch.device('cuda', 0)
CASES = ((0, 'full'), (32, 'halo prefetch from one cached line'), (64, 'no halo prefetch'), (8, 'no B stream'), (16, 'no A reads'), (24, 'no B stream, no A reads'), (2, 'no matrix phase'),
         (1, 'no split / LDS write'), (4, 'no epilogue'), (6, 'no matrix, no epilogue'), (7, 'loads + prologue only'),
         (28, 'MFMA + staging only (no B, no A, no epilogue)'))
for (C0, Cout, H, pro) in ((64, 64, 128, 1), (128, 128, 64, 1), (512, 512, 16, 1)):
    B, W = 50, H
    w = torch.randn((Cout, C0, 3, 3), device=dev) * 0.04
    pc = ops.PackedConv(w, torch.randn(Cout, device=dev), C0)
    x = torch.randn((B, H, W, C0), device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(B, C0, device=dev), 0.1 * torch.randn(B, C0, device=dev)], 1).contiguous() if pro else None
    print(f'{C0}->{Cout} @{H}x{W} B={B} prologue={pro}')
    for abl, name in CASES:
        os.environ['DMH_WINO_ABLATE'] = str(abl)
        for _ in range(3):
            ops.conv2d(pc, x, in_coef=coef, want_stats=False)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            ops.conv2d(pc, x, in_coef=coef, want_stats=False)
        e1.record()
        torch.cuda.synchronize()
        print(f'  ablate={abl:2d} {name:48s} {e0.elapsed_time(e1) * 50:8.1f} us', flush=True)

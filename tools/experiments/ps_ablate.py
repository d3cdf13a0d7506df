"""diagnostic: conv_f16x3_ps_kernel with roles ablated (needs `make -C dmhomo_amd/csrc stamps`); results are wrong by
construction, only the times matter.   python tools/ps_ablate.py
bits (DMH_WINO_ABLATE): 1 consumers skip the matrix work, 2 producers skip prologue / maximum / split / LDS write,
4 producers skip the row stores, 8 producers issue no halo loads (after the first two chunks)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('DMH_CONV_PS', '2')
from dmhomo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops

dev = torch.device('cuda', 0)
CASES = ((0, 'full'), (1, 'no matrix work'), (2, 'no staging VALU'), (3, 'no matrix, no staging VALU'), (4, 'no row stores'),
         (8, 'no halo loads'), (12, 'no loads, no stores'), (14, 'producers idle'), (15, 'everything off (barriers only)'),
         (13, 'only staging VALU'), (7, 'only halo loads'), (11, 'only row stores'))
for (C0, Cout, H, pro) in ((64, 64, 128, 1), (128, 128, 64, 1)):
    B, W = 50, H
    w = torch.randn((Cout, C0, 3, 3), device=dev) * 0.04
    pc = ops.PackedConv(w, torch.randn(Cout, device=dev), C0)
    x = torch.randn((B, H, W, C0), device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(B, C0, device=dev), 0.1 * torch.randn(B, C0, device=dev)], 1).contiguous() if pro else None
    print(f'{C0}->{Cout} @{H}x{W} B={B} prologue={pro}')
    for abl, name in CASES:
        os.environ['DMH_WINO_ABLATE'] = str(abl)
        for _ in range(3):
            ops.conv2d(pc, x, in_coef=coef, want_stats=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            ops.conv2d(pc, x, in_coef=coef, want_stats=True)
        e1.record()
        torch.cuda.synchronize()
        print(f'  ablate={abl:2d} {name:40s} {e0.elapsed_time(e1) * 50:8.1f} us', flush=True)

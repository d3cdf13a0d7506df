"""diagnostic: where the producer and the consumer waves of conv_f16x3_ps_kernel spend their cycles
(needs `make -C dmhomo_amd/csrc stamps`).    DMH_CONV_PS=2 python tools/ps_stamps.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('DMH_CONV_PS', '2')
from dmhomo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops

dev = torch.device('cuda', 0)
G = torch.cuda.get_device_properties(0).multi_processor_count & ~7
dbg = torch.zeros((G, 8, 8), dtype=torch.int64, device=dev)
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.dmh_ps_set_debug_buffer.argtypes = [ctypes.c_void_p]
raw.dmh_ps_set_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
CN = ['matrix taps A', 'matrix taps B', 'wait M', 'slab half 0', 'wait E', 'slab half 1 + wait S']
PN = ['load wait + slab -> rows', 'prologue+max', 'wait M', 'split+write, load issue', 'wait E', 'wait S']
for (C0, Cout, H, pro, k) in ((64, 64, 128, 1, 3), (128, 128, 64, 1, 3), (512, 512, 16, 1, 3), (128, 64, 128, 0, 1)):
    B, W = 50, H
    w = torch.randn((Cout, C0, k, k), device=dev) * 0.04
    pc = ops.PackedConv(w, torch.randn(Cout, device=dev), C0)
    x = torch.randn((B, H, W, C0), device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(B, C0, device=dev), 0.1 * torch.randn(B, C0, device=dev)], 1).contiguous() if pro else None
    for _ in range(3):
        ops.conv2d(pc, x, in_coef=coef, want_stats=(k == 3))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        ops.conv2d(pc, x, in_coef=coef, want_stats=(k == 3))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    d = dbg.cpu().double()
    life = d[:, :, 6]
    clk = (d[:, :, 6] / d[:, :, 7].clamp_min(1)).median().item() * 100.0
    print(f'{k}x{k} {C0}->{Cout} @{H}x{W} B={B} prologue={pro}: kernel {us:.0f} us; wave lifetime mean {life.mean():.0f} max {life.max():.0f} '
          f'cycles; in-kernel clock {clk:.0f} MHz')
    for role, names, sl in (('consumer', CN, slice(0, 4)), ('producer', PN, slice(4, 8))):
        tot = d[:, sl, 6].mean().item()
        print(f'  {role}: lifetime {tot:.0f}')
        for i in range(6):
            v = d[:, sl, i].mean().item()
            print(f'    {names[i]:34s} {v:9.0f}  {100 * v / tot:5.1f} %')

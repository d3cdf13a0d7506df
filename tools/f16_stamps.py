"""diagnostic: where a conv_f16x3 workgroup spends its cycles (needs `make -C dmhomo_amd/csrc stamps`).
    python tools/f16_stamps.py [Cin Cout H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmhomo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops

dev = torch.device('cuda', 0)
for (C0, Cout, H, pro) in ((64, 64, 128, 0), (64, 64, 128, 1), (128, 128, 64, 1), (512, 512, 16, 1)):
    B, W = 50, H
    w = torch.randn((Cout, C0, 3, 3), device=dev) * 0.04
    pc = ops.PackedConv(w, torch.randn(Cout, device=dev), C0)
    x = torch.randn((B, H, W, C0), device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(B, C0, device=dev), 0.1 * torch.randn(B, C0, device=dev)], 1).contiguous() if pro else None
    for _ in range(3):
        out, st = ops.conv2d(pc, x, in_coef=coef, want_stats=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out, st = ops.conv2d(pc, x, in_coef=coef, want_stats=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    th = 16 if Cout % 128 else 8                                  # workgroup tile height; the slot of its first stat tile
    raw = st.view(torch.int64).reshape(B, st.shape[1], -1)[:, :, :32].reshape(B, H // 8, W // 16, 4, 8)[:, ::th // 8].cpu().double()
    names = ['loads+prologue+max', 'barrier 1', 'split + LDS write', 'barrier 2', 'matrix', 'epilogue']
    tot = raw[..., 6].mean().item()
    clk = (raw[..., 6] / raw[..., 7].clamp_min(1)).median().item() * 100.0
    print(f'{C0}->{Cout} @{H}x{W} B={B} prologue={pro}: mean wave lifetime {tot:.0f} shader cycles; '
          f'in-kernel clock (s_memtime / s_memrealtime, median) {clk:.0f} MHz; kernel {us:.0f} us = '
          f'{us * clk / tot:.2f} wave lifetimes; workgroups / 512 slots = {B * (H // th) * (W // 16) * max(1, Cout // 128) / 512:.2f}')
    for i in range(6):
        print(f'  {names[i]:20s} {raw[..., i].mean().item():9.0f}  {100 * raw[..., i].mean().item() / tot:5.1f} %')

"""diagnostic (stamps build): phase stamps of the 1x1 convs that finish the up path's ResnetBlocks (cat input, no residual here)"""
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT)
from dmhomo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
for (C0, C1, Cout, H) in ((128, 64, 128, 64), (256, 128, 256, 32), (512, 256, 512, 16), (128, 0, 128, 64), (128, 64, 64, 128)):
    B, W = 50, H
    w = torch.randn((Cout, C0 + C1, 1, 1), device=dev) * 0.04
    pc = ops.PackedConv(w, torch.randn(Cout, device=dev), C0, C1)
    x = torch.randn((B, H, W, C0), device=dev)
    x1 = torch.randn((B, H, W, C1), device=dev) if C1 else None
    for _ in range(3):
        out, st = ops.conv2d(pc, x, x1, want_stats=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out, st = ops.conv2d(pc, x, x1, want_stats=True)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    th = 16 if Cout % 128 else 8
    raw = st.view(torch.int64).reshape(B, st.shape[1], -1)[:, :, :32].reshape(B, H // 8, W // 16, 4, 8)[:, ::th // 8].cpu().double()
    names = ['loads+prologue+max', 'barrier 1', 'split + LDS write', 'barrier 2', 'matrix', 'epilogue']
    if os.environ.get('DMH_CONV1_PIPE') == '1' and Cout % 128 == 0:   # (with tools/experiments/conv1x1_pipelined.patch applied)
        names = ['filling the pipe', 'scale + weight fetch issue', 'matrix', 'split', 'load issue + block maximum', 'rescale + barrier']
    tot = raw[..., 6].mean().item()
    clk = (raw[..., 6] / raw[..., 7].clamp_min(1)).median().item() * 100.0
    print(f'1x1 {C0}+{C1}->{Cout} @{H}x{W} B={B}: mean wave lifetime {tot:.0f} cycles; clock {clk:.0f} MHz; kernel {us:.0f} us = {us * clk / tot:.2f} wave lifetimes; '
          f'workgroups / 512 slots = {B * (H // th) * (W // 16) * max(1, Cout // 128) / 512:.2f}; chunks {(C0 + C1) // 32}')
    for i in range(6):
        print(f'  {names[i]:20s} {raw[..., i].mean().item():9.0f}  {100 * raw[..., i].mean().item() / tot:5.1f} %')

"""diagnostic: where a wave of the fully fused LinearAttention pass 2 spends its cycles (needs `make -C dmhomo_amd/csrc stamps`).
    python tools/la_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmhomo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
C, H, B = 64, 128, 50
x = torch.randn((B, H, H, C), device=dev)
g = torch.ones(C, device=dev)
pla = ops.PackedLinAttn(torch.randn((384, C, 1, 1), device=dev) * C ** -0.5)
plo = ops.PackedLinAttnOut(torch.randn((64, 128, 1, 1), device=dev) * 0.1, torch.zeros(64, device=dev), torch.ones(64, device=dev))
for _ in range(3):
    y = ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo)
torch.cuda.synchronize()
n = H * H
tiles = 8
nblk = (n // 64) // tiles
v = y.reshape(B, n, 64)[:, ::tiles * 64][:, :nblk, :32].reshape(B, nblk, 4, 8)[..., :6].double().cpu()
names = ['staging + q projection', 'softmax', 'split, ctx product, to_out', 'exchange write + barrier', 'LayerNorm + residual + store', 'end barrier']
tot = v.sum(-1).mean().item()
print(f'linattn_qo_kernel<true> 128x128 B=50: {tot:.0f} cycles per wave and workgroup ({tiles} sub-tiles): {tot / tiles:.0f} per sub-tile')
for i, nm in enumerate(names):
    print(f'  {nm:32s} {v[..., i].mean().item() / tiles:8.0f}  {100 * v[..., i].mean().item() / tot:5.1f} %')

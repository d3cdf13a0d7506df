"""diagnostic: where a wave of the LinearAttention pass 1 (C = 64, LDS-DMA ring) spends its cycles
(needs `make -C dmhomo_amd/csrc stamps`).     python tools/kv_stamps.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dmhomo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops
from dmhomo_amd._lib import call, ptr, lib
dev = torch.device('cuda', 0)
C, H = 64, 128
B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n = H * H
g = torch.ones(C, device=dev)
pla = ops.PackedLinAttn(torch.randn((384, C, 1, 1), device=dev) * C ** -0.5)
xs = [torch.randn((B, H, H, C), device=dev) for _ in range(4)]
ns = lib().dmh_linattn_fused_splits(B, n)
part = torch.zeros((B, ns, 4, 1088), device=dev)
for i in range(8):
    x = xs[i % 4]
    stats = torch.empty((B, n, 2), device=dev)
    call('dmh_pixel_stats', ptr(x), ptr(stats), B * n, C, 1e-5, None, 0)
    call('dmh_linattn_fused_context', ptr(x), ptr(stats), ptr(g), ptr(pla.wpack), ptr(part), B, n, C, None)
torch.cuda.synchronize()
v = part[..., :6].double().cpu()            # [B][splits][head][phase]
tiles = (n // 64) // ns
names = ['staging (raw unit -> LN -> pieces -> tile)', 'waits + barrier + DMA issue', 'fragment reads + projection MFMAs',
         'k: max, exp, sums, rescale, split p', 'v: maximum, split', 'context MFMAs + accumulate']
tot = v.sum(-1).mean().item()
print(f'linattn_kv_ring_kernel 128x128 B={B}: {tot:.0f} cycles per wave and workgroup ({tiles} sub-tiles): {tot / tiles:.0f} per sub-tile')
for i, nm in enumerate(names):
    print(f'  {nm:44s} {v[..., i].mean().item() / tiles:8.0f}  {100 * v[..., i].mean().item() / tot:5.1f} %')

"""dev tool: time the fused LinearAttention passes on the layer shapes of the BASELINE workload (HIP events).
    python tools/linattn_bench.py            (DMH_LIB_PATH=<other build> for an A/B on the same box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmhomo_amd import ops

dev = torch.device('cuda', 0)
for (C, H, fuse_out) in ((64, 128, True), (64, 64, True), (128, 64, False), (128, 32, False), (256, 32, False), (256, 16, False), (512, 16, False)):
    B = 50
    x = torch.randn((B, H, H, C), device=dev)
    g = torch.ones(C, device=dev)
    pla = ops.PackedLinAttn(torch.randn((384, C, 1, 1), device=dev) * C ** -0.5)
    plo = ops.PackedLinAttnOut(torch.randn((64, 128, 1, 1), device=dev) * 0.1, torch.zeros(64, device=dev), torch.ones(64, device=dev)) if fuse_out else None
    for _ in range(3):
        ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ops.linear_attention_fused(x, g, pla, 32 ** -0.5, out=plo)
    e1.record()
    torch.cuda.synchronize()
    print(f'linattn fused C={C:3d} {H:3d}x{H:<3d} B={B} out-fused={int(fuse_out)}: {e0.elapsed_time(e1) * 50:8.1f} us (stats + kv + merge + qo)', flush=True)

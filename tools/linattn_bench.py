"""dev tool: time the fused LinearAttention kernels on the bench shapes (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmhomo_amd import ops, _lib
lib, call, ptr = _lib.lib, _lib.call, _lib.ptr
dev = torch.device('cuda', 0)
for C, H in ((64, 128), (128, 64), (256, 32), (512, 16)):
    B, n = 50, H * H
    x = torch.randn(B, H, H, C, device=dev)
    g = torch.ones(C, device=dev)
    pla = ops.PackedLinAttn(torch.randn(384, C, 1, 1, device=dev) * C ** -0.5)
    plo = ops.PackedLinAttnOut(torch.randn(64, 128, 1, 1, device=dev) * 0.1, torch.zeros(64, device=dev), torch.ones(64, device=dev)) if C == 64 else None
    stats = torch.empty(B, n, 2, device=dev)
    ns = lib().dmh_linattn_fused_splits(B, n)
    partial = torch.empty(B, ns, 4, 1088, device=dev); ctx = torch.empty(B, 4, 32, 32, device=dev)
    out = torch.empty(B, H, H, 128, device=dev); y = torch.empty(B, H, H, 64, device=dev)
    def t(f, reps=10):
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t_st = t(lambda: call('dmh_pixel_stats', ptr(x), ptr(stats), B * n, C, 1e-5))
    t_kv = t(lambda: call('dmh_linattn_fused_context', ptr(x), ptr(stats), ptr(g), ptr(pla.wpack), ptr(partial), B, n, C))
    t_mg = t(lambda: call('dmh_linattn_merge_n', ptr(partial), ptr(ctx), B, n, ns))
    t_qo = t(lambda: call('dmh_linattn_fused_apply', ptr(x), ptr(stats), ptr(g), ptr(pla.wpack), ptr(ctx), ptr(out), B, n, C, 0.1767767))
    line = f'C={C:3d} {H}x{H}: stats {t_st:6.1f}  kv {t_kv:6.1f}  merge {t_mg:5.1f}  qo {t_qo:6.1f} us'
    if plo is not None:
        t_q2 = t(lambda: call('dmh_linattn_fused_apply_out', ptr(x), ptr(stats), ptr(g), ptr(pla.wpack), ptr(ctx), ptr(plo.wpack), ptr(plo.bias), ptr(plo.ln_g), ptr(y), B, n, C, 0.1767767, 1e-5))
        line += f'  qo+to_out+LN+res {t_q2:6.1f} us'
    print(line)

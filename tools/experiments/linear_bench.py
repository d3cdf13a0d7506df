"""time dmh_linear on the shapes of the embedding MLPs (HIP events):  python tools/experiments/linear_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
for R, i, o in [(25, 64, 256), (25, 256, 256), (25, 512, 9472), (50, 512, 9472), (25, 1, 256)]:
    x = torch.randn((R, i), device=dev); w = torch.randn((i, o), device=dev) * i ** -0.5; b = torch.randn(o, device=dev)
    y = ops.linear(x, w, b, act_in='silu')
    ref = torch.nn.functional.silu(x.double()) @ w.double() + b.double()
    err = float((y.double() - ref).abs().max())
    for _ in range(5): ops.linear(x, w, b, act_in='silu')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): ops.linear(x, w, b, act_in='silu')
    e1.record(); torch.cuda.synchronize()
    print(f'linear R={R} {i}->{o}: {e0.elapsed_time(e1) * 10:.1f} us  err {err:.1e}')

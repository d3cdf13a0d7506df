"""time of one noise draw of the headline step (25 x 6 x 128 x 128 normals) and of the class-dropout mask"""
import sys, torch
sys.path.insert(0, '/root/repo')
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
ids = torch.arange(25, dtype=torch.int64, device=dev)
st = torch.tensor([7, 0, 0, 0], dtype=torch.int64, device=dev)
for _ in range(5):
    ops.rng_indexed((25, 6, 128, 128), ids, st, 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(200):
    ops.rng_indexed((25, 6, 128, 128), ids, st, 0)
e1.record(); torch.cuda.synchronize()
print('rng_indexed normals 25x6x128x128: %.1f us per launch' % (e0.elapsed_time(e1) * 1e3 / 200))
x = torch.randn(25, 6, 128, 128, device=dev); rf = torch.randn(25, 3, 128, 128, device=dev); mk = torch.rand(25, 1, 128, 128, device=dev)
for _ in range(5):
    ops.assemble_input(x, rf, mk, reps=1, cpad=12)
torch.cuda.synchronize(); e0.record()
for _ in range(200):
    ops.assemble_input(x, rf, mk, reps=1, cpad=12)
e1.record(); torch.cuda.synchronize()
print('assemble_input 25 rows 128x128 -> 12 channels NHWC: %.1f us per launch' % (e0.elapsed_time(e1) * 1e3 / 200))
from dmhomo_amd.ops import call, ptr, lib
for (B, n) in ((25, 16384), (25, 4096), (25, 1024)):
    ns = lib().dmh_linattn_fused_splits(B, n)
    partial = torch.randn(B, ns, 4, 1088, device=dev).abs()
    ctx = torch.empty(B, 4, 32, 32, device=dev)
    for _ in range(5):
        call('dmh_linattn_merge_n', ptr(partial), ptr(ctx), B, n, ns, None)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        call('dmh_linattn_merge_n', ptr(partial), ptr(ctx), B, n, ns, None)
    e1.record(); torch.cuda.synchronize()
    print('linattn_merge B=%d n=%d (%d splits): %.1f us per launch' % (B, n, ns, e0.elapsed_time(e1) * 1e3 / 200))

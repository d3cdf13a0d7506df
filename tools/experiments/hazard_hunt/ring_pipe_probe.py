"""where do the launches of the pipelined ring variant of pass 1 (DMH_LA_RING=2) differ from the rows computed alone when a
conv runs beside them?  prints the count / magnitude / position of the differing partials of pass 1 itself"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from gpu_util import rand
from dmhomo_amd import ops
from dmhomo_amd._lib import call, ptr, lib
dev = torch.device('cuda', 0)
C, H, B = 64, 128, 25
g = (1 + 0.2 * rand((C,), 51)).to(dev)
pla = ops.PackedLinAttn(rand((384, C, 1, 1), 52, C ** -0.5).to(dev))
x = (rand((B, H, H, C), 50) * 1.3 + 0.2).to(dev)
n = H * H
stats = torch.empty((B, n, 2), device=dev)
call('dmh_pixel_stats', ptr(x), ptr(stats), B * n, C, 1e-5, None, 0)
ns = lib().dmh_linattn_fused_splits(B, n)
def kv(xx, ss):
    b = xx.shape[0]
    part = torch.zeros((b, ns, 4, 1088), device=dev)
    call('dmh_linattn_fused_context', ptr(xx), ptr(ss), ptr(g), ptr(pla.wpack), ptr(part), b, n, C, None)
    return part
alone = kv(x[:2].contiguous(), stats[:2].contiguous())
w = rand((128, 128, 3, 3), 90, (1.0 / 1152) ** 0.5).to(dev)
pc = ops.PackedConv(w, None, 128)
xc = rand((25, 64, 64, 128), 91).to(dev)
side = torch.cuda.Stream(device=dev)
for it in range(8):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            ops.conv2d(pc, xc)
    p = kv(x, stats)
    torch.cuda.current_stream().wait_stream(side)
    d = (p[:2] - alone).abs()
    bad = (d > 0)
    if bad.any():
        idx = bad.nonzero()
        splits = sorted(set((int(i[0]), int(i[1])) for i in idx))
        heads = sorted(set(int(i[2]) for i in idx))
        cols = idx[:, 3]
        print(it, 'differing floats', int(bad.sum()), 'max', float(d.max()), 'rel', float((d / alone.abs().clamp_min(1e-20)).max()),
              'splits', splits[:6], 'n', len(splits), 'heads', heads, 'in max/sum/ctx', int((cols < 32).sum()), int(((cols >= 32) & (cols < 64)).sum()), int((cols >= 64).sum()))
    else:
        print(it, 'equal')

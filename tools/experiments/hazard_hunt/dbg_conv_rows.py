import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from gpu_util import rand
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
for (name, C0, C1, Co, H, k) in (('3x3 64->64 128', 64, 0, 64, 128, 3), ('3x3 64+64->64 128', 64, 64, 64, 128, 3), ('3x3 256->256 32', 256, 0, 256, 32, 3), ('1x1 64+64->64 128', 64, 64, 64, 128, 1), ('3x3 512->512 16', 512, 0, 512, 16, 3)):
    B = 50
    w = rand((Co, C0 + C1, k, k), 1, (1.0 / ((C0 + C1) * k * k)) ** 0.5).to(dev)
    pc = ops.PackedConv(w, rand((Co,), 2, 0.1).to(dev), C0, C1)
    x0 = rand((B, H, H, C0), 3).to(dev)
    x1 = rand((B, H, H, C1), 4).to(dev) if C1 else None
    coef = torch.stack([1 + 0.1 * rand((B, C0), 5), 0.1 * rand((B, C0), 6)], 1).contiguous().to(dev) if (C1 == 0 and k == 3) else None
    res = rand((B, H, H, Co), 7).to(dev)
    def run(n):
        o = ops.conv2d(pc, x0[:n].contiguous(), None if x1 is None else x1[:n].contiguous(), in_coef=None if coef is None else coef[:n].contiguous(), res=res[:n].contiguous(), want_stats=(k == 3))
        return o if isinstance(o, tuple) else (o, None)
    a, sa = run(2)
    ok = True
    for _ in range(3):
        b, sb = run(B)
        ok = ok and bool(torch.equal(b[:2], a)) and (sa is None or bool(torch.equal(sb[:2], sa)))
    print(name, 'rows of B=50 equal the rows alone (3 launches):', ok)

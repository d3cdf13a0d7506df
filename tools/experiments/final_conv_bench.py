import sys, torch
sys.path.insert(0, '/root/repo')
from dmhomo_amd import ops
dev = torch.device('cuda', 0)
x = torch.randn((25, 128, 128, 64), device=dev); w = torch.randn((6, 64), device=dev) * 0.1; b = torch.randn(6, device=dev)
ref = torch.einsum('bhwc,oc->bohw', x.double(), w.double()) + b.double()[None, :, None, None]
out = ops.final_conv_nchw(x, w, b)
print('err', float((out.double() - ref).abs().max()))
for _ in range(5): ops.final_conv_nchw(x, w, b)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): ops.final_conv_nchw(x, w, b)
e1.record(); torch.cuda.synchronize()
print('final_conv 25x128x128x64->6: %.1f us' % (e0.elapsed_time(e1) * 1e3 / 50))

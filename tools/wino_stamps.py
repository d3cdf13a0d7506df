"""diagnostic: where a Winograd conv workgroup spends its cycles (needs `make -C dmhomo_amd/csrc stamps`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['DMH_CONV3_VARIANT'] = '6'
from dmhomo_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'dmhomo_amd', 'libdmhomo_hip_stamps.so')
import torch
from dmhomo_amd import ops

dev = torch.device('cuda', 0)
B, H, W, C0, Cout = 50, 128, 128, 64, 64
if len(sys.argv) > 1:
    C0 = Cout = int(sys.argv[1]); H = W = int(sys.argv[2])
w = torch.randn((Cout, C0, 3, 3), device=dev) * 0.04
pc = ops.PackedConv(w, torch.randn(Cout, device=dev), C0)
x = torch.randn((B, H, W, C0), device=dev)
for _ in range(3):
    out, st = ops.conv2d(pc, x, want_stats=True)
torch.cuda.synchronize()
tiles = st.shape[1]
raw = st.view(torch.int64).reshape(B, tiles, -1)[:, :, :32].reshape(B, tiles, 4, 8).cpu().double()
names = ['raw write', 'barrier 1', 'transform', 'barrier 2', 'matrix', 'epilogue', 'total', 'start']
tot = raw[..., 6].mean().item()
print(f'shape {C0}->{Cout} @{H}x{W}, B={B}: mean wave lifetime {tot:.0f} cycles (memtime ticks)')
for i in range(6):
    print(f'  {names[i]:12s} {raw[..., i].mean().item():9.0f}  {100 * raw[..., i].mean().item() / tot:5.1f} %')
regs = st.view(torch.int64).reshape(B, tiles, -1)[:, :, :32].reshape(B, tiles, 4, 8)[..., 7].cpu().numpy().ravel()
import collections
lds_alloc = (regs >> 32) & 0xffffffff
hw_id = regs & 0xffffffff
print('  LDS_ALLOC values:', collections.Counter(int(v) for v in lds_alloc).most_common(6))
print('  LDS_BASE  [7:0]  :', collections.Counter(int(v) & 0xff for v in lds_alloc).most_common(6))
print('  HW_ID wave_id[3:0]:', collections.Counter(int(v) & 0xf for v in hw_id).most_common(8))
print('  HW_ID tg_id[19:16]:', collections.Counter((int(v) >> 16) & 0xf for v in hw_id).most_common(8))

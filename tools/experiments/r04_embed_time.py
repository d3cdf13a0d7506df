"""times of the launches at the head of a CFG pass (25 rows): embeddings, small linears, the (scale, shift) linear"""
import sys, torch
sys.path.insert(0, '/root/repo')
from dmhomo_amd import cfg, ops
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1).to(dev)
eng = m._engine
eng.ensure_prepared()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
t = torch.full((B,), 500, dtype=torch.int64, device=dev)
c = torch.zeros(B, dtype=torch.int64, device=dev)
k = torch.ones(B, dtype=torch.uint8, device=dev)
cond = eng.embed(t, [(c, k)], 1)
print('mlp_wt', tuple(eng.mlp_wt.shape))
def timeit(name, f, n=200):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) * 1e3 / n:.1f} us')
timeit('embed (2 embeds + 4 linears)', lambda: eng.embed(t, [(c, k)], 1))
timeit('scale_shift linear 512 -> %d' % eng.mlp_wt.shape[1], lambda: ops.linear(cond, eng.mlp_wt, eng.mlp_b, act_in='silu'))
se = ops.sinusoidal_embed(t, eng.freq)
timeit('linear 64 -> 256 gelu', lambda: ops.linear(se, eng.t_w1, eng.t_b1, act_out='gelu'))

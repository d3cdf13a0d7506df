"""Round 6 probe: would BALANCING the de-duplicated rows over the two streams pay?  cfg.Unet.dedup_dropped_rows runs the null pass
(25 rows) on one stream and the kept conditional rows (~12 of 25) on the other; this times — as captured graphs, the way the
sampling step runs them — two UNet trunks on two streams over the 2B-row layout with FIXED row lists of
    (25, 12)   today's split (null rows | kept conditional rows)
    (19, 18)   the same 37 rows dealt evenly
    (37,)      one stream, one launch sequence
    (25, 25)   the full 2B rows (the headline's two passes), for scale.
The row lists are arbitrary subsets (timing only: the results of a list that mixes passes are not the sampler's).
    python tools/experiments/dedup_balance_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmhomo_amd import cfg, ddpm, ops

dev = torch.device('cuda', 0)
torch.manual_seed(0)
B, S = 25, 128
m = cfg.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=6, num_classes=1).to(dev)
eng = m._engine
eng.ensure_prepared()
data, classes = next(ddpm.SyntheticConditions(S, B, seed=1000, device=dev))
rgb_flow, mask = data[:, -5:-2].contiguous(), data[:, -6:-5].contiguous()
x = torch.randn((B, 6, S, S), device=dev)
t = torch.full((B,), 500, device=dev, dtype=torch.long)
x0 = m._stem(x, ops.affine(rgb_flow, 2., -1.), mask)
x02 = x0.repeat(2, 1, 1, 1)
keep1 = torch.ones(B, dtype=torch.uint8, device=dev)
null = torch.zeros(B, dtype=torch.uint8, device=dev)
cond = eng.embed(t, [(classes, keep1), (classes, null)], 2)
streams = (torch.cuda.Stream(), torch.cuda.Stream())


def rows_list(slots):
    return torch.tensor([len(slots)] + list(slots) + [0] * (2 * B - len(slots)), dtype=torch.int32, device=dev)


def body(lists):
    cur = torch.cuda.current_stream()
    for st, rl in zip(streams, lists):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            eng.trunk(x02, cond, rows=rl)
    for st in streams[:len(lists)]:
        cur.wait_stream(st)


cond_c, cond_n = cond[:B].contiguous(), cond[B:].contiguous()
TODAY_LISTS = {k: torch.tensor([k] + list(range(k)) + [0] * (B - k), dtype=torch.int32, device=dev) for k in (12, 25)}


def body_today(n_kept):
    """the product's own layout: two B-row trunks, the conditional one on a list of its kept rows"""
    cur = torch.cuda.current_stream()
    for st, (cnd, r_) in zip(streams, ((cond_n, None), (cond_c, TODAY_LISTS[n_kept] if n_kept < B else None))):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            eng.trunk(x0, cnd, rows=r_)
    for st in streams:
        cur.wait_stream(st)


def timed(lists, reps=24):
    run = (lambda: body_today(lists)) if isinstance(lists, int) else (lambda: body(lists))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        run()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


allr = list(range(2 * B))
cases = [('product: B-row grids, 25 | 12 kept', 12), ('product: B-row grids, 25 | 25 (headline)', 25), ('today  (25 | 12)', [allr[B:], allr[:12]]), ('even   (19 | 18)', [allr[B:B + 19], allr[B + 19:] + allr[:12]]),
         ('one stream (37)', [allr[B:] + allr[:12]]), ('full   (25 | 25)', [allr[B:], allr[:B]]),
         ('even, 3 streams is not tried: two streams (13 | 12 | 12 as 25 | 12 above)', None)]
for rnd in range(2):
    for name, slots in cases:
        if slots is None:
            continue
        arg = slots if isinstance(slots, int) else [rows_list(s_) for s_ in slots]
        torch.cuda.synchronize()
        ms = timed(arg)
        print(f'{name:42s} {ms:7.3f} ms per two-trunk forward (2B-row grids, graph replay) [{rnd}]', flush=True)

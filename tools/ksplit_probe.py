"""Round 6, VERDICT r5 "Next" item 1: what would a SHAPE-FIXED split-K of the <= 32^2 convolutions deliver?  Measured by an exact
emulation with the product kernel, before building anything:

  a conv Cin -> Cout on B rows split ks ways along K  ==  (workgroup for workgroup: same count, same chunks per workgroup, same
  weight and input traffic, same full-tile fp32 store per workgroup)  a conv (Cin / ks) -> Cout on B * ks rows,

plus the second kernel a split needs: the fixed-order sum of the ks partial tensors (+ bias, + GroupNorm partials) — timed here
as a plain fp32 sum over the leading axis (ks reads + 1 write per element: its HBM floor; a fused reduce cannot be cheaper).
The headline runs the two CFG passes as two 25-row launch sequences on two streams, so every case is timed three ways:
  alone        one launch of B rows                       (the regime of a step whose streams do not meet at this layer)
  two streams  the same launch on two streams at once     (the regime when they do: 2B rows of workgroups on the chip)
  dedup        25 rows on one stream + 12 on the other    (cfg.Unet.dedup_dropped_rows' average)
    python tools/ksplit_probe.py [--reps 30]
Prints one table per shape: us for ks = 1 / 2 / 4 (conv alone, conv + reduce), workgroups and rounds of 512 slots."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmhomo_amd import ops

SHAPES = [  # name, H, Cin (C0 + C1 of the launch), Cout
    ('512->512 @16^2 (downs.3 / mid blocks)', 16, 512, 512),
    ('512+512->512 @16^2 (ups.0 block1)', 16, 1024, 512),
    ('256->256 @32^2 (downs.2)', 32, 256, 256),
    ('256+256->256 @32^2 (ups.1 block1)', 32, 512, 256),
    ('128->128 @64^2 (downs.1; for scale)', 64, 128, 128),
]


def make(B, H, cin, cout, dev):
    w = torch.randn((cout, cin, 3, 3), device=dev) * (1.0 / (cin * 9)) ** 0.5
    pc = ops.PackedConv(w, torch.randn(cout, device=dev), cin)
    x = torch.randn((B, H, H, cin), device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(B, cin, device=dev), 0.1 * torch.randn(B, cin, device=dev)], 1).contiguous()
    bound = torch.full((B, 8), 32.0, device=dev)
    return pc, x, coef, bound


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, H, cin, cout in SHAPES:
        print(f'== {name}')
        print(f'{"regime":>12} {"ks":>2} {"workgroups":>10} {"rounds/512":>10} {"conv us":>8} {"reduce us":>9} {"total us":>8} {"vs ks=1":>7}')
        wg_per_row = (H // 8) * (H // 16) * max(1, cout // 128)
        for regime, rows in (('alone', (25,)), ('two streams', (25, 25)), ('dedup', (25, 12)), ('batched', (50,))):
            base = None
            for ks in (1, 2, 4):
                if (cin // ks) % 32:
                    continue
                sets = [make(r * ks, H, cin // ks, cout, dev) for r in rows]
                parts = [torch.empty((ks, r, H, H, cout), device=dev) for r in rows]
                outs = [torch.empty((r, H, H, cout), device=dev) for r in rows]

                def conv_all():
                    cur = torch.cuda.current_stream()
                    if len(sets) == 1:
                        pc, x, coef, bound = sets[0]
                        ops.conv2d(pc, x, in_coef=coef, in_bound=bound, want_stats=(ks == 1))
                        return
                    for st, (pc, x, coef, bound) in zip((s1, s2), sets):
                        st.wait_stream(cur)
                        with torch.cuda.stream(st):
                            ops.conv2d(pc, x, in_coef=coef, in_bound=bound, want_stats=(ks == 1))
                    cur.wait_stream(s1)
                    cur.wait_stream(s2)

                def reduce_all():
                    if ks == 1:
                        return
                    cur = torch.cuda.current_stream()
                    if len(sets) == 1:
                        torch.sum(parts[0], dim=0, out=outs[0])
                        return
                    for st, p_, o_ in zip((s1, s2), parts, outs):
                        st.wait_stream(cur)
                        with torch.cuda.stream(st):
                            torch.sum(p_, dim=0, out=o_)
                    cur.wait_stream(s1)
                    cur.wait_stream(s2)
                tc = timed(conv_all, a.reps)
                tr = timed(reduce_all, a.reps) if ks > 1 else 0.0
                both = timed(lambda: (conv_all(), reduce_all()), a.reps)
                base = base or both
                wgs = sum(rows) * wg_per_row * ks
                print(f'{regime:>12} {ks:2d} {wgs:10d} {wgs / 512:10.2f} {tc:8.1f} {tr:9.1f} {both:8.1f} {both / base:7.2f}', flush=True)
                del sets, parts, outs
        print()


if __name__ == '__main__':
    main()

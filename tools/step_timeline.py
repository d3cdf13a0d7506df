"""analysis behind tools/step_timeline.sh: over the last 31 mid-steps of the last sample() call of a kernel trace, the time
with k kernels running, and for the time with exactly ONE kernel running which kernel it was (a step's serial section and
its under-filled stretches)."""
import csv, sys
from collections import Counter, defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seek = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('sampler_seek')]
t0, t1 = int(rows[seek[-32]]['End_Timestamp']), int(rows[seek[-2]]['End_Timestamp'])
ks = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
ev = []
for i, r in enumerate(ks):
    ev.append((int(r['Start_Timestamp']), 1, i))
    ev.append((int(r['End_Timestamp']), -1, i))
ev.sort()
def fam(n):
    n = n.split('(')[0]
    for key in ('conv_f16x3_kernel<3, 3', 'conv_f16x3_kernel<1, 1', 'conv_f16x3_kernel<7', 'conv_f16x3_kernel<2, 2', 'linattn_qo', 'linattn_kv',
                'gn_finalize', 'gn_silu_residual', 'linattn_merge', 'linear_kernel', 'chan_layernorm', 'pixel_stats', 'attention_kernel',
                'rng_', 'sampler_', 'assemble', 'ss_gather', 'class_embed', 'sinusoidal'):
        if key in n:
            return key
    return n[:40]
running, last = set(), t0
by_depth, solo = Counter(), defaultdict(int)
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        by_depth[min(len(running), 3)] += dt
        if len(running) == 1:
            solo[fam(ks[next(iter(running))]['Kernel_Name'])] += dt
    if d > 0: running.add(i)
    else: running.discard(i)
    last = t
span = t1 - t0
print(f'{len(ks)} launches over {span / 31e3:.1f} us per denoise step (31 steps)')
for k in sorted(by_depth): print(f'  {k}{"+" if k == 3 else ""} kernels running: {by_depth[k] / 31e3:8.1f} us per step = {100 * by_depth[k] / span:5.1f} %')
print('  exactly one kernel running, by family (us per step):')
for f, v in sorted(solo.items(), key=lambda kv: -kv[1])[:16]: print(f'    {v / 31e3:8.1f}  {f}')

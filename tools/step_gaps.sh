# dev tool: GPU idle time between the replays of the per-step HIP graph (gap between the cursor kernel that ends a denoise step and
# the next kernel) and the fraction of a sampling call during which some kernel runs, from a rocprofv3 kernel trace of bench.py.
#   gpurun -- bash tools/step_gaps.sh      (round 3: 4.6 us between steps, a kernel running 99.2 % of the time)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/gaps && mkdir -p $R/gpurun_out/gaps
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps -o g -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/gaps/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/gaps/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seek = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('sampler_seek')]
gaps = []
for i in seek:
    if i + 1 < len(rows):
        end = max(int(r['End_Timestamp']) for r in rows[max(0, i - 3):i + 1])
        gaps.append((int(rows[i + 1]['Start_Timestamp']) - end, rows[i + 1]['Kernel_Name'][:40]))
gaps = gaps[-40:]
print('gap between the cursor kernel that ends a denoise step and the next kernel (us):')
print(' '.join(f'{g / 1e3:.1f}' for g, _ in gaps))
print('next kernels:', sorted(set(n for _, n in gaps)))
# whole-run busy fraction over the last sample() call
t0, t1 = int(rows[seek[-33]]['End_Timestamp']), int(rows[seek[-1]]['End_Timestamp'])
inside = [r for r in rows if t0 <= int(r['Start_Timestamp']) <= t1]
ev = sorted([(int(r['Start_Timestamp']), 1) for r in inside] + [(int(r['End_Timestamp']), -1) for r in inside])
busy, depth, last = 0, 0, t0
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
print(f'last 32 steps: {(t1 - t0) / 1e6:.2f} ms, some kernel running {100 * busy / (t1 - t0):.2f} % of it')
PY

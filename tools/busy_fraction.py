"""dev tool: GPU busy fraction of a rocprofv3 --kernel-trace CSV (union of kernel intervals / span), overall and for the
last `--tail` fraction of the run (the timed steps)."""
import argparse, csv
ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--tail', type=float, default=0.5)
a = ap.parse_args()
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(a.trace)))
iv = iv[int(len(iv) * (1 - a.tail)):]
busy, cur_s, cur_e, total_k = 0, iv[0][0], iv[0][1], 0
for s, e in iv:
    total_k += e - s
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0]
print(f'kernels {len(iv)}  span {span/1e6:.1f} ms  busy(union) {busy/1e6:.1f} ms = {busy/span:.3f}  sum of durations {total_k/1e6:.1f} ms '
      f'(overlap factor {total_k/busy:.2f})')

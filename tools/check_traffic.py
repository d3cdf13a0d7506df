"""fails (exit 1) when a fresh PMC traffic figure deviates from the committed one by more than a tolerance.

    python tools/check_traffic.py <fresh pmc_traffic.json> <committed profiles/r0N_pmc_traffic.json> [tolerance, default 0.03]

bench.py's `roofline.traffic` is READ from the committed file (the default bench run cannot collect counters: rocprofv3 --pmc
needs its own passes), so a traffic regression would not show in the driver's line; the round's profile script
(tools/profile_round5.sh) runs this check on every fresh pass instead."""
import json
import sys

fresh, committed = (json.load(open(p)) for p in sys.argv[1:3])
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
bad = []
for key in ('hbm_bytes_per_launch', 'fetch_bytes_per_launch', 'write_bytes_per_launch'):
    a, b = fresh[key], committed[key]
    dev = abs(a - b) / b
    print(f'{key}: fresh {a / 1e6:.2f} MB, committed {b / 1e6:.2f} MB, deviation {dev * 100:.2f} %')
    if key == 'hbm_bytes_per_launch' and dev > tol:
        bad.append(key)
if fresh.get('kernel') != committed.get('kernel'):
    bad.append(f"kernel filter changed: {fresh.get('kernel')!r} vs {committed.get('kernel')!r}")
if bad:
    print(f'TRAFFIC CHECK FAILED (> {tol * 100:.0f} %): {bad}')
    sys.exit(1)
print('traffic check ok')

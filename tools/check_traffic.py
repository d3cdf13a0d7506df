"""fails (exit 1) when a fresh PMC traffic figure deviates from the committed one by more than a tolerance.

    python tools/check_traffic.py <fresh pmc_traffic.json> <committed profiles/rNN_pmc_traffic.json | profiles/> [tolerance, default 0.03]

A DIRECTORY as the second argument means "the newest committed figure in it" (the highest rNN_pmc_traffic.json) — the file
bench.py's `roofline.traffic_committed` quotes (bench.newest_committed_traffic resolves it the same way), so the round's profile
script and the bench line always compare against the same baseline.  Since round 5 bench.py measures `roofline.traffic` itself
(two rocprofv3 --pmc child passes after the timed region) and prints `traffic_vs_committed`; this check is the same comparison
for the profile script's own passes (tools/profile_round6.sh)."""
import glob
import json
import os
import sys

if os.path.isdir(sys.argv[2]):
    sys.argv[2] = sorted(glob.glob(os.path.join(sys.argv[2], 'r[0-9][0-9]_pmc_traffic.json')))[-1]
    print('committed figure:', sys.argv[2])
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

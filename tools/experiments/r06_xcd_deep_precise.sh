# a tighter A/B of DMH_CONV_XCD_DEEP (0 / 16 / 32): 10 alternating rounds, 8 timed steps each
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6xcd; mkdir -p $O; cd $R
for r in 1 2 3 4 5 6 7 8 9 10; do
  for k in 0 32 16; do
    v=$(DMH_CONV_XCD_DEEP=$k python3 bench.py --steps 8 --warmup 2 --no-variants --no-traffic --no-cpu-baseline --no-roofline --no-phases 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "$r $k $v"
  done
done > $O/ab.txt
python3 - <<PY
import collections, statistics
d = collections.defaultdict(list)
for l in open('$O/ab.txt'):
    r, k, v = l.split(); d[k].append(float(v))
for k, v in d.items():
    print(k, 'mean %.3f  median %.3f  stdev %.3f  n %d' % (statistics.mean(v), statistics.median(v), statistics.stdev(v), len(v)))
base = d['0']
for k in ('32', '16'):
    diffs = [a / b - 1 for a, b in zip(d[k], base)]
    print('XCD_DEEP=%s vs 0: paired mean %+.3f %%  (stderr %.3f %%)' % (k, 100 * statistics.mean(diffs), 100 * statistics.stdev(diffs) / len(diffs) ** 0.5))
PY

# generic tight A/B of environment settings on the headline step: bash tools/experiments/r06_ab.sh "A=1 B=2" "A=0" ... (first = baseline)
# 10 alternating rounds, 8 timed steps each; paired differences against the first setting
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6ab; mkdir -p $O; cd $R
rm -f $O/ab.txt
ROUNDS=${ROUNDS:-10}
for r in $(seq 1 $ROUNDS); do
  i=0
  for cfg in "$@"; do
    v=$(env $cfg python3 bench.py --steps 8 --warmup 2 --no-variants --no-traffic --no-cpu-baseline --no-roofline --no-phases $EXTRA 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "$r $i $v" >> $O/ab.txt
    i=$((i+1))
  done
done
python3 - "$@" <<PY
import collections, statistics, sys
names = sys.argv[1:]
d = collections.defaultdict(list)
for l in open('$O/ab.txt'):
    r, k, v = l.split(); d[int(k)].append(float(v))
for k in sorted(d):
    v = d[k]
    print('%-48s mean %.3f  median %.3f  stdev %.3f  n %d' % (names[k], statistics.mean(v), statistics.median(v), statistics.stdev(v), len(v)))
for k in sorted(d):
    if k == 0: continue
    diffs = [a / b - 1 for a, b in zip(d[k], d[0])]
    print('%-48s vs baseline: paired mean %+.3f %%  (stderr %.3f %%)' % (names[k], 100 * statistics.mean(diffs), 100 * statistics.stdev(diffs) / len(diffs) ** 0.5))
PY

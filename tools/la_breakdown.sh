# per-kernel durations of the fused LinearAttention block on the bench shapes: gpurun -- bash tools/la_breakdown.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/la
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/la -o l -- python3 $R/tools/linattn_bench.py > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/la/**/l_kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0][:34]
    agg[(n, r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    if len(v) >= 20:
        print(k, len(v), round(sum(v[3:]) / len(v[3:]), 1), 'us')
PY

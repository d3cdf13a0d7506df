# usage: bash tools/pmc_conv.sh <shape substring>   (rocprofv3 PMC pass over tools/conv_bench.py, one shape)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv
rm -rf $OUT
rocprofv3 --kernel-trace --pmc ${PMC:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA} --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/tools/conv_bench.py --only $1 --reps 5 > $OUT.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
    n[(k, r['Counter_Name'])] += 1
for k, c in agg.items():
    if 'conv' not in k or 'pack' in k: continue
    print(k)
    for name, v in c.items(): print(f'   {name:28s} {v / n[(k, name)]:16.0f} per launch ({n[(k, name)]})')
PY

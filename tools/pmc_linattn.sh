# rocprofv3 PMC evidence for the fused LinearAttention at the 128x128 level (C = 64, B = 50 rows: linattn_kv_kernel<2>,
# linattn_qo_kernel<true>): matrix-pipe share, clock, HBM bytes.  Every counter pass is its own run with --kernel-trace only.
#   gpurun -- bash tools/pmc_linattn.sh ; then  cp gpurun_out/pmc_la/r03_pmc_linattn.json profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_la
rm -rf $O && mkdir -p $O
B="python3 $R/tools/linattn_bench.py --only 0"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/sq -o p -- $B > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/time -o p -- $B > $O/time.log 2>&1
python3 - $O <<'PY'
import collections, csv, glob, json, sys
O = sys.argv[1]
KERNELS = {'pass 1 (kv)': 'linattn_kv_ring_kernel', 'pass 2 (qo + to_out + LN + residual)': 'linattn_qo_kernel<true>'}
GRID = "409600"          # 128x128, B = 50: 1600 workgroups
def counters(d, kern):
    f = glob.glob(O + '/' + d + '/**/*counter_collection.csv', recursive=True)[0]
    agg, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name'] and r.get('Grid_Size', r.get('Grid_Size_X', GRID)) in (GRID, '1600'):
            agg[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    return {k: v / n[k] for k, v in agg.items()}, (max(n.values()) if n else 0)
def avg_us(kern):
    f = glob.glob(O + '/time/**/*kernel_trace.csv', recursive=True)[0]
    v = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(f))
         if kern in r['Kernel_Name'] and r['Grid_Size_X'] == GRID]
    return sum(v[3:]) / len(v[3:]), len(v)
res = {'shape': 'C = 64, 128x128, B = 50 rows (tools/linattn_bench.py), fused to_out', 'kernels': {}}
alg = {'pass 1 (kv)': 50 * 16384 * 64 * 4, 'pass 2 (qo + to_out + LN + residual)': 2 * 50 * 16384 * 64 * 4}
for name, kern in KERNELS.items():
    sq, n = counters('sq', kern); fe, _ = counters('fetch', kern); wr, _ = counters('write', kern)
    us, calls = avg_us(kern)
    d = {'kernel': kern, 'avg_launch_us': us, 'launches_timed': calls, 'counters_per_launch': sq}
    if sq:
        d['effective_clock_GHz'] = sq.get('GRBM_GUI_ACTIVE', 0) / 8 / (us * 1e-6) / 1e9
        d['mfma_busy_frac_of_SIMD_cycles'] = sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (sq.get('GRBM_GUI_ACTIVE', 1) / 8 * 1024)
    if fe and wr:
        hb = fe.get('FETCH_SIZE', 0) * 1024 * 2 + wr.get('WRITE_SIZE', 0) * 1024     # KiB; FETCH_SIZE x2 on gfx950
        d['hbm_bytes_per_launch'] = hb
        d['algorithmic_bytes_per_launch'] = alg[name]
        d['traffic_over_algorithmic'] = hb / alg[name]
        d['algorithmic_GB/s'] = alg[name] / us / 1e3
        d['frac_of_8TB/s'] = alg[name] / (us * 1e-6) / 8e12
    res['kernels'][name] = d
json.dump(res, open(O + '/pmc_linattn.json', 'w'), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY

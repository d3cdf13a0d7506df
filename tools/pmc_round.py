"""summarise the passes of tools/pmc_round.sh -> one JSON (copied into profiles/ by hand).
    python tools/pmc_round.py <pass dir> <out.json>
Per-launch averages over the launches of the canonical shape's kernel (the prologue variant AND the no-prologue variant of
tools/conv_bench.py's `3x3_64_64_128*` rows run the same instantiation; both are included and counted)."""
import collections, csv, glob, json, os, sys

KERNEL = 'conv_f16x3_kernel<3, 3, 1, 0, 16, 16, 4, 1>'


def counters(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        return {}, 0
    agg, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if KERNEL in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            n[r['Counter_Name']] += 1
    return {k: v / n[k] for k, v in agg.items()}, (max(n.values()) if n else 0)


def avg_us(d):
    f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)
    if not f:
        return None
    for r in csv.DictReader(open(f[0])):
        if KERNEL in r['Name']:
            return float(r['AverageNs']) / 1e3, int(r['Calls'])
    return None


def main():
    O, out = sys.argv[1], sys.argv[2]
    res = {'kernel': KERNEL, 'shape': '3x3 64->64 @128x128, B = 50 rows (tools/conv_bench.py --only 3x3_64_64_128: with and without '
           'the GN+SiLU prologue), GroupNorm partials on', 'passes': {}}
    for name in ('sq_random', 'sq_zeros', 'sq_mops', 'ta', 'cache', 'fetch', 'write'):
        c, n = counters(os.path.join(O, name))
        res['passes'][name] = {'launches': n, 'per_launch': c}
    tr, tz = avg_us(os.path.join(O, 'time_random')), avg_us(os.path.join(O, 'time_zeros'))
    res['avg_launch_us'] = {'random_operands': tr and tr[0], 'zero_operands': tz and tz[0],
                            'note': 'rocprofv3 --kernel-trace --stats (no counters): same instruction stream on all-zero activations '
                                    'and weights vs random ones; a large ratio = the chip holds its clock down under the fp16 MFMA load'}
    if tr and tz:
        res['avg_launch_us']['random_over_zero'] = tr[0] / tz[0]
    d = {}
    r = res['passes']['sq_random']['per_launch']
    if r and tr:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3 (MI355X_MICROARCH.md, DVFS give-back)
        clk = r.get('GRBM_GUI_ACTIVE', 0) / 8 / (tr[0] * 1e-6) / 1e9
        d['effective_clock_GHz'] = clk
        d['mfma_busy_frac_of_SIMD_cycles'] = (r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) /
                                              (r.get('GRBM_GUI_ACTIVE', 1) / 8 * 1024))      # 256 CUs x 4 SIMDs
        d['mfma_instructions_per_launch'] = r.get('SQ_INSTS_MFMA')
        d['busy_cycles_per_mfma'] = r.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(r.get('SQ_INSTS_MFMA', 1), 1)
    z = res['passes']['sq_zeros']['per_launch']
    if z and tz:
        d['effective_clock_GHz_zero_operands'] = z.get('GRBM_GUI_ACTIVE', 0) / 8 / (tz[0] * 1e-6) / 1e9
    f, w = res['passes']['fetch']['per_launch'], res['passes']['write']['per_launch']
    if f and w:
        # counters are in KiB; FETCH_SIZE counts 64 B per 128 B request on gfx950 -> x2 (MI355X_MICROARCH.md, HBM)
        fb, wb = 2 * f.get('FETCH_SIZE', 0) * 1024, w.get('WRITE_SIZE', 0) * 1024
        d['hbm_bytes_per_launch'] = {'fetch_x2': fb, 'write': wb, 'total': fb + wb,
                                     'algorithmic': 4.0 * 50 * 128 * 128 * 128 + 4 * (9 * 64 * 64 + 3 * 64) + 8 * 50 * 64}
    res['derived'] = d
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res['derived'], indent=1))
    print(json.dumps(res['avg_launch_us'], indent=1))


if __name__ == '__main__':
    main()

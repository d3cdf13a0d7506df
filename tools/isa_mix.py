"""dev tool: instruction mix of one kernel in `hipcc -S` output, priced with the SIMD issue rates measured by
tools/micro/valu_rate.hip (gfx950: full rate 2.4 cycles per wave64 instruction, half 4.2, quarter 8.2; MFMA 16x16x32 16.3):
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only dmhomo_amd/csrc/linattn_fused.hip -o /tmp/la.s
    python tools/isa_mix.py /tmp/la.s linattn_kv_ring_kernel [--top 25]
Static counts (an unrolled loop body counts once, a rolled one too): read them per loop trip."""
import argparse, collections, re

QUARTER = ('v_exp_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_log_', 'v_sin_', 'v_cos_', 'v_fma_mixlo', 'v_fma_mixhi', 'v_permlane')
HALF = ('v_pk_', 'v_cvt_', 'v_max_f32', 'v_min_f32', 'v_max3_f32', 'v_min3_f32', 'v_fma_mix_f32', 'v_perm_b32', 'v_med3')


def rate(op, text):
    if op.startswith('v_mfma'):
        return 'mfma', 16.3 if '16x16x32' in op else (32.5 if '32x32x16' in op else 16.3)
    if op.startswith(QUARTER):
        return 'quarter', 8.2
    if op.startswith(HALF) or '_dpp' in op or 'quad_perm' in text or 'row_' in text or '_sdwa' in op:
        return 'half', 4.2
    if op.startswith('v_'):
        return 'full', 2.4
    if op.startswith('ds_'):
        return 'lds', 0.0
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
        return 'vmem', 0.0
    if op.startswith('s_'):
        return 'scalar', 0.0
    return 'other', 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('asm')
    ap.add_argument('kernel', help='substring of the (mangled) kernel name')
    ap.add_argument('--top', type=int, default=20)
    a = ap.parse_args()
    lines = open(a.asm).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^[_A-Za-z0-9]+:', l) and a.kernel in l)
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    ops, cls, cyc = collections.Counter(), collections.Counter(), collections.Counter()
    for l in lines[start + 1:end]:
        l = l.split(';')[0].strip()
        if not l or l.startswith('.') or l.endswith(':'):
            continue
        op = l.split()[0]
        c, r = rate(op, l)
        ops[op] += 1
        cls[c] += 1
        cyc[c] += r
    print(f'{lines[start][:-1]}: {sum(ops.values())} instructions')
    for c in ('mfma', 'full', 'half', 'quarter', 'lds', 'vmem', 'scalar', 'other'):
        print(f'  {c:8s} {cls[c]:6d}  {cyc[c]:9.0f} SIMD cycles')
    valu = cyc['full'] + cyc['half'] + cyc['quarter']
    print(f'  vector-ALU cycles / MFMA cycles = {valu:.0f} / {cyc["mfma"]:.0f}')
    for op, n in ops.most_common(a.top):
        print(f'    {op:34s} {n}')


if __name__ == '__main__':
    main()

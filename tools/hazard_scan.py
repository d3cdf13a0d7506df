#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for hazards hipcc's recognizer cannot see because one side is inline asm.

Inline asm shows up between ;;#ASMSTART / ;;#ASMEND.  For every VALU instruction inside such a block report, within WINDOW
preceding instructions:
  RAW   a source register written by an MFMA (vDst) fewer than passes + 4 wait states earlier (see need())
  WAW   the destination written by an MFMA
  WAR-C the destination read as SrcC by an MFMA whose vDst is a different register range
and for every MFMA: A/B/C sources written by an inline-asm VALU less than NOPS wait states earlier;
  TRANS a source written by a transcendental instruction (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos) in the
        instruction directly in front (gfx940+ forwards a transcendental result one wait state late; hipcc pads its own
        consumers, not an inline-asm one).
  ASM-MFMA (round 3: conv_f16x3.hip issues its MFMAs from inline asm, three per accumulator) for every vector / LDS / vector-memory
        instruction OUTSIDE inline asm: any register it names that is the vDst of an inline-asm MFMA fewer than passes + 4 wait
        states earlier — hipcc does not know those statements are MFMAs and pads nothing behind them (MFMA -> MFMA on the same
        accumulator is interlocked by the hardware and not reported).
Run by tests/test_isa_hazards.py over every source that contains inline-asm vector instructions.
"""
import re, sys
WINDOW = 20   # the longest wait the gfx950 tables ask for (16-pass MFMA result -> VALU)
def need(op):
    """wait states between an MFMA and a VALU that reads / overwrites its vDst (or overwrites its SrcC) on gfx950: passes + 4
    (LLVM GCNHazardRecognizer, GFX940_XDL_N_PassWriteVgprVALU*WaitStates with the gfx950 extra state).  Passes from the
    measured issue cycles of MI355X_MICROARCH.md (4 cycles per pass): 16x16x32 16-bit -> 4, 32x32x16 16-bit and 16x16x4 f32
    -> 8, 32x32x2 f32 -> 16; anything unknown is treated as 16 passes."""
    if '16x16x32' in op or '16x16x16' in op or '4x4' in op: return 8
    if '32x32x16' in op or '32x32x8' in op or '16x16x4' in op: return 12
    return 20
def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'-?\|?v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'-?\|?v(\d+)\b', tok)
    if m: return {int(m.group(1))}
    return set()
def scan(path):
    kern = None; hist = []; in_asm = False; n_find = 0
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        if s.endswith(':') and not s.startswith(';') and not s.startswith('.L'):
            kern = s[:-1]; hist = []; continue
        if s.startswith(';;#ASMSTART'): in_asm = True; continue
        if s.startswith(';;#ASMEND'): in_asm = False; continue
        if not s or s[0] in ';.': continue
        body = s.split(';')[0].strip()
        parts = body.split(None, 1)
        op = parts[0]; ops = parts[1].split(',') if len(parts) > 1 else []
        ops = [o.strip() for o in ops]
        waits = 1
        if op == 's_nop': waits = int(ops[0]) + 1
        if op.startswith('v_mfma'):
            d, a, b, c = regs(ops[0]), regs(ops[1]), regs(ops[2]), regs(ops[3])
            # producers in asm
            dist = 0
            for h in reversed(hist):
                if h['asm'] and h['dst'] & (a | b | c) and dist < 4:
                    print(f'{kern}:{ln}: MFMA reads v{sorted(h["dst"] & (a|b|c))} written by inline asm {dist} wait states earlier (line {h["ln"]})'); n_find += 1
                dist += h['waits']
                if dist > WINDOW: break
            hist.append(dict(kind='mfma', dst=d, c=c, asm=False, in_asm=in_asm, ln=ln, waits=1, need=need(op)))
        elif in_asm and op.startswith('v_'):
            d = regs(ops[0]); srcs = set()
            for o in ops[1:]: srcs |= regs(o)
            if 'mixhi' in op or 'mixlo' in op: srcs |= d
            if hist and hist[-1].get('trans') and hist[-1]['dst'] & srcs:
                print(f'{kern}:{ln}: asm {op} reads v{sorted(hist[-1]["dst"] & srcs)} written by the transcendental instruction directly in front (line {hist[-1]["ln"]})'); n_find += 1
            dist = 0
            for h in reversed(hist):
                if h['kind'] == 'mfma' and dist < h['need']:
                    if h['dst'] & srcs: print(f'{kern}:{ln}: asm {op} RAW on MFMA vDst v{sorted(h["dst"] & srcs)} ({dist} wait states, mfma line {h["ln"]})'); n_find += 1
                    if h['dst'] & d: print(f'{kern}:{ln}: asm {op} WAW on MFMA vDst v{sorted(h["dst"] & d)} ({dist} wait states, mfma line {h["ln"]})'); n_find += 1
                    if h['c'] & d and h['c'] != h['dst']: print(f'{kern}:{ln}: asm {op} WAR on MFMA SrcC v{sorted(h["c"] & d)} ({dist} wait states, mfma line {h["ln"]})'); n_find += 1
                dist += h['waits']
                if dist > WINDOW: break
            hist.append(dict(kind='valu', dst=d, asm=True, ln=ln, waits=1))
        else:
            if not in_asm and op.startswith(('v_', 'ds_', 'global_', 'buffer_', 'scratch_', 'flat_')):
                named = set()
                for o in ops: named |= regs(o)
                dist = 0
                for h in reversed(hist):
                    if h['kind'] == 'mfma' and h.get('in_asm') and dist < h['need'] and h['dst'] & named:
                        print(f'{kern}:{ln}: {op} names v{sorted(h["dst"] & named)}, the vDst of an inline-asm MFMA {dist} wait states earlier (line {h["ln"]})'); n_find += 1
                    dist += h['waits']
                    if dist > WINDOW: break
            d = regs(ops[0]) if ops and op.startswith(('v_', 'ds_read', 'global_load', 'buffer_load')) else set()
            trans = (not in_asm) and op.split('_e')[0] in ('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32')
            hist.append(dict(kind='other', dst=d, asm=False, ln=ln, waits=waits, trans=trans))
        if len(hist) > 4 * WINDOW: hist = hist[-2 * WINDOW:]
    return n_find
# ---- LDS-DMA rings (round 4).  A kernel that fills LDS with global_load_lds_* and hands the data to OTHER waves through a
# raw s_barrier is right only if, at every barrier, every piece of the unit about to be read has landed — and a wave can
# only vouch for its OWN pieces, by an s_waitcnt vmcnt in front of ITS arrival at the barrier.  Checked on the generated
# code, per kernel that contains an LDS-DMA instruction, by a forward dataflow over its control-flow graph:
#   state   = upper bound of the vector-memory operations this wave may still have in flight (every VMEM instruction adds
#             one — vmcnt counts them all —, s_waitcnt vmcnt(N) caps it at N; joins take the maximum; loops to a fixpoint)
#   DMA-BAR   at an s_barrier the bound must be 0 — or, at the FIRST barrier of the kernel (the priming of a ring that keeps
#             one unit ahead), at most `ahead` operations, all of them issued after the unit that is consumed first
#   DMA-M0    global_load_lds_* takes its LDS base from M0: the s_mov_b32 m0 must be the last SALU write in front of it with
#             at least one wait state (s_nop) between, and nothing else may write M0 in between
VMEM = ('global_load', 'global_store', 'global_atomic', 'buffer_load', 'buffer_store', 'buffer_atomic', 'flat_load', 'flat_store',
        'flat_atomic', 'scratch_load', 'scratch_store')
def scan_lds_dma(path, ahead=5, verbose=False):
    kernels, cur = {}, None
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        head = s.split(';')[0].strip()
        if head.endswith(':') and not head.startswith('.'):
            cur = head[:-1]; kernels[cur] = []; continue
        if cur is None or not s or s[0] == ';': continue
        if s.startswith('.Lfunc_end'):
            cur = None; continue                            # (blocks may follow the first s_endpgm: the function ends here)
        if s.startswith('.L') and s.split(';')[0].strip().endswith(':'):
            kernels[cur].append((ln, 'label', s.split(':')[0])); continue
        if s[0] == '.': continue
        body = s.split(';')[0].strip()
        if not body: continue
        kernels[cur].append((ln, 'ins', body))
    n_find = 0
    for kern, ins in kernels.items():
        if not any(k == 'ins' and b.startswith('global_load_lds') for _, k, b in ins):
            continue
        # basic blocks
        blocks, cur_b = [], {'labels': [], 'ins': []}
        for ln, k, b in ins:
            if k == 'label':
                if cur_b['ins'] or cur_b['labels']:
                    blocks.append(cur_b)
                cur_b = {'labels': [b], 'ins': []}
                continue
            cur_b['ins'].append((ln, b))
            op = b.split()[0]
            if op.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc')):
                blocks.append(cur_b); cur_b = {'labels': [], 'ins': []}
        if cur_b['ins'] or cur_b['labels']: blocks.append(cur_b)
        label_of = {}
        for i, b in enumerate(blocks):
            for l in b['labels']: label_of[l] = i
        succ = []
        for i, b in enumerate(blocks):
            out = []
            last = b['ins'][-1][1] if b['ins'] else ''
            op = last.split()[0] if last else ''
            tgt = last.split()[-1] if last else ''
            if op.startswith('s_branch'): out = [label_of[tgt]]
            elif op.startswith('s_cbranch'): out = [label_of[tgt]] + ([i + 1] if i + 1 < len(blocks) else [])
            elif op.startswith(('s_endpgm', 's_setpc')): out = []
            elif i + 1 < len(blocks): out = [i + 1]
            succ.append(out)
        CAP = 63
        def transfer(state, b, report):
            nonlocal n_find
            o, first = state
            m0_age = None          # instructions since the last s_mov_b32 m0 (None: M0 not set in this block yet)
            for ln, body in b['ins']:
                op = body.split()[0]
                if op == 's_mov_b32' and body.split()[1].rstrip(',') == 'm0':
                    m0_age = 0
                elif op.startswith('global_load_lds'):
                    if report and (m0_age is None or m0_age < 1):
                        print(f'{kern}:{ln}: DMA-M0 {op} without s_mov_b32 m0 + a wait state directly in front'); n_find += 1
                    o = min(o + 1, CAP)
                    m0_age = None
                elif op.startswith(VMEM):
                    o = min(o + 1, CAP)
                elif op == 's_waitcnt':
                    m = re.search(r'vmcnt\((\d+)\)', body)
                    if m: o = min(o, int(m.group(1)))
                elif op == 's_barrier':
                    allowed = ahead if first else 0
                    if report and o > allowed:
                        print(f'{kern}:{ln}: DMA-BAR s_barrier reached with up to {o} vector-memory operations of this wave in '
                              f'flight (allowed {allowed}): another wave may read LDS-DMA data that has not landed'); n_find += 1
                    if report and verbose: print(f'{kern}:{ln}: s_barrier, in flight <= {o} (first={first})')
                    first = False
                if m0_age is not None and not (op == 's_mov_b32' and body.split()[1].rstrip(',') == 'm0'):
                    if op.startswith('s_') and not op.startswith('s_nop') and 'm0' in body.split(None, 1)[-1].split(',')[0]:
                        m0_age = None      # M0 written by something else
                    else:
                        m0_age += 1
            return (o, first)
        IN = [None] * len(blocks)
        IN[0] = (0, True)
        work = [0]
        while work:
            i = work.pop()
            out = transfer(IN[i], blocks[i], False)
            for j in succ[i]:
                new = out if IN[j] is None else (max(IN[j][0], out[0]), IN[j][1] and out[1])
                if new != IN[j]:
                    IN[j] = new; work.append(j)
        for i, b in enumerate(blocks):
            if IN[i] is not None: transfer(IN[i], b, True)
    return n_find
if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    tot = sum(scan(p) for p in args) + sum(scan_lds_dma(p, verbose='--verbose' in sys.argv) for p in args)
    print('findings:', tot)

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
if __name__ == '__main__':
    tot = sum(scan(p) for p in sys.argv[1:])
    print('findings:', tot)

#!/usr/bin/env python
"""Static count of the instructions in the loops of a gfx950 assembly listing (hipcc -S --cuda-device-only): for every kernel and
every backward branch, the instructions between the target label and the branch, by class -- VALU full rate, VALU quarter rate
(v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos), packed fp32 (v_pk_*_f32: two lanes' worth per slot), VMEM, LDS, SALU --
and `slots` = full + 4 x quarter (the issue slots of a wave64 on a 16-lane SIMD are 4 clocks each; a quarter-rate operation
occupies 4 of them).  `per` divides by the number of v_exp_f32 in the body (one per logit in the gamma = 2 loss kernels).
No GPU needed: this is how a change to an arithmetic-bound kernel is sized before it is measured.

    python tools/isa_loops.py file.s [substring of the kernel name] [--blocks] [--quarter-cost=4]

--quarter-cost: issue slots charged per quarter-rate instruction (default 4, the architectural rate; MI355X_MICROARCH.md measures
the ISSUE cost of a transcendental beside other work at ~5/3 of a plain VALU operation -- pass 1.67 to count that way).

--blocks: instead of the loops, every straight-line block (label / branch to label / branch) that holds a v_exp_f32 -- the
arithmetic of one code path without the other paths of the same loop mixed in.
"""
import re
import sys

QUARTER = ('v_exp_', 'v_log_', 'v_rcp_', 'v_rsq_', 'v_sqrt_', 'v_sin_', 'v_cos_')


def classify(op):
    if op.startswith('v_'):
        if op.startswith(QUARTER):
            return 'quarter'
        if op.startswith('v_pk_') and op.endswith('_f32'):
            return 'packed'
        if op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')):
            return 'valu'
        return 'valu'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def kernels(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r'^(_Z\w+):', line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur and line.strip().startswith('.end_amdhsa_kernel'):
            cur = None
        if cur and re.match(r'^\s*s_endpgm', line):
            body.append(line)
            yield cur, body
            cur = None
            continue
        if cur:
            body.append(line)


def loops(body):
    labels, ins = {}, []
    for line in body:
        s = line.split(';')[0].strip()
        if not s or s.startswith('.') and not s.endswith(':'):
            continue
        m = re.match(r'^(\.?\w+):$', s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        ins.append(s)
    for i, s in enumerate(ins):
        m = re.match(r'^s_c?branch\w*\s+(\.?\w+)', s)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            yield m.group(1), ins[labels[m.group(1)]:i + 1]


def blocks(body):
    label, seq = 'entry', []
    for line in body:
        s = line.split(';')[0].strip()
        if not s or s.startswith('.') and not s.endswith(':'):
            continue
        m = re.match(r'^(\.?\w+):$', s)
        if m:
            if seq:
                yield label, seq
            label, seq = m.group(1), []
            continue
        seq.append(s)
        if re.match(r'^s_c?branch|^s_endpgm|^s_setpc', s):
            yield label, seq
            label, seq = label + '+', []
    if seq:
        yield label, seq


def main():
    text = open(sys.argv[1]).read()
    args = [a for a in sys.argv[2:] if not a.startswith('--')]
    want = args[0] if args else ''
    by_block = '--blocks' in sys.argv
    quarter_cost = next((float(a.split('=', 1)[1]) for a in sys.argv[2:] if a.startswith('--quarter-cost=')), 4.0)
    for name, body in kernels(text):
        if want not in name:
            continue
        print(name)
        for label, seq in (blocks(body) if by_block else loops(body)):
            if by_block and not any(s.startswith('v_exp_f32') for s in seq):
                continue
            n = {'valu': 0, 'quarter': 0, 'packed': 0, 'vmem': 0, 'lds': 0, 'salu': 0, 'other': 0}
            for s in seq:
                n[classify(s.split()[0])] += 1
            exps = sum(1 for s in seq if s.startswith('v_exp_f32'))
            slots = n['valu'] + n['packed'] + quarter_cost * n['quarter']
            per = ' = %.1f slots per v_exp' % (slots / exps) if exps else ''
            print('  %-4s %-10s %5d instr: valu %4d  packed %3d  quarter %3d  vmem %3d  lds %2d  salu %3d  -> %.0f VALU slots%s'
                  % ('block' if by_block else 'loop', label, len(seq), n['valu'], n['packed'], n['quarter'], n['vmem'], n['lds'], n['salu'], slots, per))


if __name__ == '__main__':
    main()

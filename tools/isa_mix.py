#!/usr/bin/env python
"""Instruction mix of the time loop of a kernel in a hipcc -S listing:  python tools/isa_mix.py file.s <substring of the kernel symbol>
The loop is taken to be the longest backward branch of the function."""
import collections
import re
import sys

src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
i0 = next(i for i, l in enumerate(src) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
i1 = next(i for i in range(i0, len(src)) if src[i].startswith('.Lfunc_end'))
body = src[i0:i1]
labels = {l.split(':')[0]: k for k, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
best = (0, 0, 0)
for k, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < k and k - labels[m.group(1)] > best[0]:
        best = (k - labels[m.group(1)], labels[m.group(1)], k)
loop = body[best[1]:best[2] + 1]
cnt = collections.Counter()
for l in loop:
    m = re.match(r'^\s+([a-z_0-9]+)\s', l + ' ')
    if m and not l.strip().startswith(('.', ';')):
        cnt[m.group(1)] += 1


def cls(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op in ('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32'):
        return 'trans'
    if re.match(r'v_.*_f64|v_cvt_f64|v_cvt_f32_f64', op):
        return 'valu_f64'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')):
        return 'vmem'
    return 'other'


c2 = collections.Counter()
for op, n in cnt.items():
    c2[cls(op)] += n
print('function lines', len(body), 'loop lines', len(loop), 'instructions', sum(cnt.values()))
print(dict(c2))
for op, n in cnt.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f'{op:30s}{n}')

#!/usr/bin/env python
"""Where in a kernel's time loop does the compiler spill?  python tools/spill_map.py file.s <kernel symbol substring> [bucket]
Histogram of scratch stores (S) / loads (L) against the running count of MFMA instructions (the step is fully unrolled, so
the MFMA count identifies the phase: see the table in DESIGN.md section 4.1)."""
import collections
import sys
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 50
i0 = next(i for i, l in enumerate(src) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
i1 = next(i for i in range(i0, len(src)) if src[i].startswith('.Lfunc_end'))
m, ev = 0, []
for l in src[i0:i1]:
    if 'v_mfma' in l:
        m += 1
    if 'scratch_store' in l:
        ev.append(('S', m))
    if 'scratch_load' in l:
        ev.append(('L', m))
h = collections.Counter((e[0], e[1] // bucket * bucket) for e in ev)
for b in sorted(set(k[1] for k in h)):
    print(f"mfma {b:5d}..{b + bucket - 1:5d}   stores {h.get(('S', b), 0):3d}   loads {h.get(('L', b), 0):3d}")
print('total mfma', m, 'stores', sum(v for k, v in h.items() if k[0] == 'S'), 'loads', sum(v for k, v in h.items() if k[0] == 'L'))

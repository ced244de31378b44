#!/bin/bash
# Register / scratch use of every kernel in one HIP source:  tools/kres.sh <file.hip> [extra hipcc flags]
# (hipcc -Rpass-analysis=kernel-resource-usage, one line per kernel)
F=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-pass-failed "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/kres.o $F 2>&1 | python3 -c "
import re,sys
cur={}
for l in sys.stdin:
    m=re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPR Spill|SGPR Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (.*?) \[-R', l)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k=='Function Name':
        if cur: print(cur)
        cur={'fn':v}
    else: cur[k.split(' [')[0]]=v
if cur: print(cur)
"

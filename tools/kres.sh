#!/bin/bash
# kernel resource usage of one translation unit: tools/kres.sh <unit.hip> [extra flags]  -> name, VGPRs, AGPRs, scratch, occupancy, LDS
C=/root/repo/gs-2m_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-slp-vectorize -std=c++17 -I$C $2 -Rpass-analysis=kernel-resource-usage -c $C/$1 -o /tmp/kres.o 2>&1 | python3 -c "
import sys, re
cur = {}
for l in sys.stdin:
    m = re.search(r'remark:\s+(.*?)\s+\[-Rpass', l)
    if not m: continue
    t = m.group(1).strip()
    k, _, v = t.partition(':')
    k = k.strip(); v = v.strip()
    if k == 'Function Name':
        cur = {'name': v}
    else:
        cur[k] = v
        if k.startswith('LDS Size'):
            n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', cur['name'])[:48]
            print('%-50s VGPR %-4s AGPR %-3s scratch %-5s spill %-3s occ %-3s LDS %s' % (n, cur.get('VGPRs'), cur.get('AGPRs'), cur.get('ScratchSize [bytes/lane]'), cur.get('VGPRs Spill'), cur.get('Occupancy [waves/SIMD]'), v))
"

#!/usr/bin/env python3
"""Static instruction mix of one kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only): tools/isa_mix.py file.s kernel_substring"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(txt) if re.match(r'^_Z.*' + re.escape(sys.argv[2]) + r'.*:', l)][0]
fe = [i for i, l in enumerate(txt) if i > start and l.startswith('.Lfunc_end')][0]
lines = [l.strip() for l in txt[start + 1:fe] if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.strip().endswith(':')]
c = Counter()
for l in lines:
    op = l.split()[0]
    if op.startswith('v_') and 'dpp' in l: c['valu_dpp:' + op] += 1
    elif op.startswith('v_'): c['valu:' + op] += 1
    elif op.startswith('s_'): c['salu:' + op] += 1
    elif op.startswith('ds_'): c['lds:' + op] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem:' + op] += 1
    else: c[op] += 1
tot = lambda p: sum(v for k, v in c.items() if k.startswith(p))
print(len(lines), 'instructions: valu', tot('valu'), 'salu', tot('salu'), 'lds', tot('lds'), 'vmem', tot('vmem'))
for k, v in sorted(c.items(), key=lambda x: -x[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print('  %-40s %d' % (k, v))

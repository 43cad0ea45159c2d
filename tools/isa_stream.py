#!/usr/bin/env python3
"""Condensed instruction stream of one loop of a kernel in a hipcc -S listing: M = MFMA, r = ds_read, w = ds_write, G = global
load, . = VALU, , = SALU, [..] = s_waitcnt, |B| = barrier.   Usage: isa_stream.py file.s <mangled-substring> <.LBBx_y>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key, lab = sys.argv[2], sys.argv[3]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
j = next(i for i, l in enumerate(body) if l.startswith(lab + ":"))
out = []
while j < len(body):
    x = body[j].strip(); j += 1
    if not x or x[0] in '.;':
        continue
    op = x.split()[0]
    if op.startswith('v_mfma'): out.append('M')
    elif op.startswith('ds_read'): out.append('r')
    elif op.startswith('ds_write'): out.append('w')
    elif op.startswith('s_waitcnt'): out.append('[' + x.split(None, 1)[1].replace('lgkmcnt', 'l').replace('vmcnt', 'v') + ']')
    elif op.startswith(('global_load', 'buffer_load')): out.append('G')
    elif op.startswith(('global_store', 'buffer_store')): out.append('S')
    elif op.startswith('scratch_'): out.append('X')
    elif op.startswith('s_barrier'): out.append('|B|')
    elif op.startswith('v_'): out.append('.')
    elif op.startswith(('s_cbranch', 's_branch')):
        out.append('<br>')
        if lab in x: break
    else: out.append(',')
print(''.join(out))

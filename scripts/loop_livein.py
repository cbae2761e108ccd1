#!/usr/bin/env python3
"""Development aid: live-in registers and instruction mix of the largest loop of a kernel in a
gfx950 .s file:  loop_livein.py file.s mangled-kernel-name-substring"""
import re, sys, collections
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); i = s.index(':', i); j = s.index('.Lfunc_end', i)
lines = s[i:j].split('\n')
labels = {}
for n, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = n
loops = []
for n, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < n: loops.append((n - labels[m.group(1)], labels[m.group(1)], n))
loops = [x for x in loops if 'Loop Header' in lines[x[1]]] or loops
loops.sort(reverse=True)
_, a, b = loops[0]
loop = lines[a:b + 1]
written, livein = set(), set()
def regs(tok):
    out = []
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]', tok):
        out += [(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)]
    for m in re.finditer(r'(?<![\[:\w])([va])(\d+)\b', tok):
        out.append((m.group(1), int(m.group(2))))
    return out
for l in loop:
    l = l.split(';')[0].strip()
    if not l or l.endswith(':') or l.startswith('.'): continue
    parts = l.split(None, 1)
    if len(parts) < 2: continue
    ops = [o.strip() for o in parts[1].split(',')]
    op = parts[0]
    if op.startswith(('scratch_store', 'global_store', 'ds_write', 's_', 'buffer_store')): dst, src = [], ops
    else: dst, src = ops[:1], ops[1:]
    if op.startswith('v_fmac'): src = ops
    for o in src:
        for r in regs(o):
            if r not in written: livein.add(r)
    for o in dst:
        for r in regs(o): written.add(r)
c = collections.Counter(l.split()[0] for l in loop if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':'))
print('loop lines', len(loop), 'live-in VGPR', len([r for r in livein if r[0] == 'v']), 'AGPR', len([r for r in livein if r[0] == 'a']))
print('f64', sum(v for k, v in c.items() if 'f64' in k), 'accvgpr', c['v_accvgpr_read_b32'] + c['v_accvgpr_write_b32'] + c.get('v_accvgpr_mov_b32', 0),
      'other valu', sum(v for k, v in c.items() if k.startswith('v_') and 'f64' not in k and 'accvgpr' not in k),
      'lds', sum(v for k, v in c.items() if k.startswith('ds_')), 'scratch', sum(v for k, v in c.items() if k.startswith('scratch')),
      'vmem', sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_'))), 'salu', sum(v for k, v in c.items() if k.startswith('s_')))

"""Instruction census per basic block of one kernel in a --save-temps .s file.  Usage: isa_sections.py <file.s> <mangled-substring>"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^\w*%s\w*:' % pat, l)]
i = start[0]; j = i
while 's_endpgm' not in lines[j]: j += 1
sec = []; cur = ['entry', {}]
for ln in lines[i + 1:j + 1]:
    ln = ln.strip()
    if not ln or ln.startswith(';'): continue
    if ln.startswith('.LBB'):
        sec.append(cur); cur = [ln.split()[0], {}]; continue
    if ln.startswith('.') or ln.endswith(':'): continue
    op = ln.split()[0]
    cur[1][op] = cur[1].get(op, 0) + 1
sec.append(cur)
for name, c in sec:
    tot = sum(c.values()); v = sum(n for o, n in c.items() if o.startswith('v_'))
    if tot > 20:
        top = sorted(c.items(), key=lambda kv: -kv[1])[:9]
        print(name, 'total', tot, 'valu', v, top)

#!/usr/bin/env python3
"""Development tool: print the instructions of one kernel whose debug location (innermost line of FILE in the inlined-at
chain, from a -gline-tables-only -save-temps .s) falls in [LO, HI].
usage: isa_region.py FILE.s KERNEL_SUBSTRING SRCFILE LO HI [occurrence]"""
import sys, re
path, kern, src, lo, hi = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
occ = int(sys.argv[6]) if len(sys.argv) > 6 else 0
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and kern in l.split(':')[0] and ':' in l)
cur = None; out = []; runs = []; inrun = False
for l in lines[start:]:
    t = l.strip()
    if t.startswith('.Lfunc_end'): break
    if t.startswith('.loc'):
        chain = re.findall(re.escape(src) + r':(\d+)', t)
        cur = int(chain[0]) if chain else None
        continue
    if not t or t[0] in ';.' and not t.startswith('.LBB'): continue
    if t.startswith(';;'): continue
    hit = cur is not None and lo <= cur <= hi
    if t.endswith(':') or t.startswith('.LBB'):
        if inrun: runs[-1].append('      ' + t)
        continue
    if hit:
        if not inrun: runs.append([]); inrun = True
        runs[-1].append('%4d  %s' % (cur, t.split(';')[0].strip()))
    else:
        if inrun and len(runs[-1]) < 3: pass
        inrun = False
big = [r for r in runs if len(r) >= 8]
print('%d runs, %d big' % (len(runs), len(big)))
r = big[occ] if occ < len(big) else []
print('\n'.join(r))

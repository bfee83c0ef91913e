#!/usr/bin/env python3
"""Development tool: static instruction counts of one kernel per source line, from a -gline-tables-only -save-temps .s file.
usage: isa_by_line.py FILE.s KERNEL_SUBSTRING [source-file-substring]"""
import sys, re, collections
path, kern = sys.argv[1], sys.argv[2]
srcsel = sys.argv[3] if len(sys.argv) > 3 else None
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and kern in l and ':' in l.split()[0])
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
cur = (0, 0)
valu = collections.Counter(); salu = collections.Counter(); other = collections.Counter()
for l in lines[start:]:
    t = l.strip()
    if t.startswith('.loc'):
        p = t.split()
        cur = (int(p[1]), int(p[2]))
        continue
    if t.startswith('.Lfunc_end'): break
    if not t or t[0] in ';.' or t.endswith(':'): continue
    op = t.split()[0]
    if op.startswith('v_'): valu[cur] += 1
    elif op.startswith('s_'): salu[cur] += 1
    else: other[cur] += 1
keys = sorted(set(valu) | set(salu) | set(other))
tot = [0, 0, 0]
for k in keys:
    f = files.get(k[0], '?')
    if srcsel and srcsel not in f: 
        pass
    print("%-28s:%-5d valu %4d salu %4d other %3d" % (f.split('/')[-1], k[1], valu[k], salu[k], other[k]))
    tot[0] += valu[k]; tot[1] += salu[k]; tot[2] += other[k]
print("total valu %d salu %d other %d" % tuple(tot))

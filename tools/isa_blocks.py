#!/usr/bin/env python3
"""Development tool: per basic block of one kernel in a -save-temps .s: instruction count, VALU count, SGPR-spill traffic
(v_readlane / v_writelane), scratch accesses, and the loop depth the compiler notes.
usage: isa_blocks.py FILE.s MANGLED_KERNEL_PREFIX [min_instructions]"""
import sys, re, collections
path, name = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(name) and ':' in l.split()[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
blk = 'entry'; stats = collections.OrderedDict(); stats[blk] = [0, 0, 0, 0, 0, '']
for l in lines[start + 1:end]:
    t = l.strip()
    if t.startswith('.LBB') or t.startswith('; %bb'):
        blk = t.split(';')[0].strip() or t
        m = re.search(r'Depth=(\d+)', t)
        stats[blk] = [0, 0, 0, 0, 0, 'depth ' + m.group(1) if m else '']
        continue
    if not t or t[0] in '.;': continue
    op = t.split()[0]
    s = stats[blk]
    s[0] += 1
    if op.startswith('v_'): s[1] += 1
    if op == 'v_readlane_b32': s[2] += 1
    if op == 'v_writelane_b32': s[3] += 1
    if op.startswith('scratch_'): s[4] += 1
tot = [0] * 5
for b, s in stats.items():
    for i in range(5): tot[i] += s[i]
    if s[0] >= minn: print('%-28s n %4d valu %4d readlane %3d writelane %3d scratch %2d %s' % (b[:28], s[0], s[1], s[2], s[3], s[4], s[5]))
print('total n %d valu %d readlane %d writelane %d scratch %d' % tuple(tot))

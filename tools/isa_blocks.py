#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing.  usage: tools/isa_blocks.py <file.s> <symbol-prefix>"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
i = next(k for k, l in enumerate(lines) if l.startswith(pref) and ':' in l)
end = next(j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end'))
stats, order, bb = {}, [], None
for l in lines[i + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m or bb is None:
        bb = m.group(1) if m else 'entry'
        order.append(bb)
        stats[bb] = {'v': 0, 's': 0, 'smem': 0, 'vmem': 0, 'lds': 0, 'br': 0}
        if m:
            continue
    t = l.strip().split(' ')[0] if l.strip() else ''
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    if t.startswith('v_'):
        stats[bb]['v'] += 1
    elif t.startswith('s_load') or t.startswith('s_buffer'):
        stats[bb]['smem'] += 1
    elif t.startswith('s_cbranch') or t.startswith('s_branch'):
        stats[bb]['br'] += 1
    elif t.startswith('s_'):
        stats[bb]['s'] += 1
    elif t.startswith('global_') or t.startswith('buffer_') or t.startswith('flat_'):
        stats[bb]['vmem'] += 1
    elif t.startswith('ds_'):
        stats[bb]['lds'] += 1
tot = {k: sum(s[k] for s in stats.values()) for k in ('v', 's', 'smem', 'vmem', 'lds', 'br')}
print('total', tot)
for b in order:
    print(b, stats[b])

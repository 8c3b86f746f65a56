"""Compressed event map of one kernel's ISA (device-only .s): M = MFMA, g = global load, A = global atomic, D = LDS-DMA,
S / L = scratch store / load, w = s_waitcnt vmcnt, : = label, | = branch.  usage: isa_map.py file.s <kernel substring>"""
import itertools, re, sys
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(want) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
ev = []
for l in lines[start:end]:
    t = l.strip()
    if t.startswith('v_mfma'): ev.append('M')
    elif t.startswith('scratch_store') or (t.startswith('buffer_store') and 'offen' not in t and 's[0:3]' in t): ev.append('S')
    elif t.startswith('scratch_load') or (t.startswith('buffer_load') and 's[0:3]' in t): ev.append('L')
    elif t.startswith('global_atomic'): ev.append('A')
    elif t.startswith('global_load_lds'): ev.append('D')
    elif t.startswith('global_load'): ev.append('g')
    elif t.startswith('global_store'): ev.append('s')
    elif t.startswith('s_waitcnt') and 'vmcnt' in t: ev.append('w')
    elif t.startswith('s_cbranch') or t.startswith('s_branch'): ev.append('|')
    elif t.startswith('.LBB'): ev.append(':')
print(end - start, 'lines')
print(' '.join(k + (str(n) if n > 1 else '') for k, n in ((k, len(list(g))) for k, g in itertools.groupby(ev))))

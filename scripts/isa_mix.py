"""Instruction mix of the basic blocks of one kernel in a hipcc -S listing: python scripts/isa_mix.py file.s <symbol prefix> [min MFMAs]"""
import re, sys, collections
s = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
start = next(i for i, l in enumerate(s) if l.startswith(pref) and l.rstrip().split(':')[0].startswith(pref) and ':' in l)
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
body = s[start:end]
lab = [i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)] + [len(body)]
def cls(op):
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith('v_'): return 'VALU'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith('s_waitcnt'): return 'WAIT'
    if op.startswith('s_'): return 'SALU'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'VMEM'
    return 'other'
for a, b in zip(lab[:-1], lab[1:]):
    ops = [l.strip().split()[0] for l in body[a:b] if l.strip() and not l.strip().startswith((';', '.'))]
    c = collections.Counter(cls(o) for o in ops)
    if c['MFMA'] >= minm:
        print(body[a], dict(c))
        print('  VALU:', collections.Counter(o for o in ops if cls(o) == 'VALU').most_common(30))
        print('  LDS :', collections.Counter(o for o in ops if cls(o) == 'LDS').most_common())

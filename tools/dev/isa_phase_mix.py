"""Dev tool: instruction mix between consecutive s_barrier instructions of a kernel in an ISA listing.
    python tools/dev/isa_phase_mix.py file.s <substring of mangled name>"""
import re, sys, collections
sys.path.insert(0, __file__.rsplit('/', 2)[0])
def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'lane'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    return 'other'
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and pat in l and ':' in l and '@' in l)
end = start
while not lines[end].startswith('.Lfunc_end'):
    end += 1
c = collections.Counter(); seg = 0; ops = collections.Counter()
for l in lines[start:end]:
    b = l.strip()
    if not b or b[0] in ';.' or b.endswith(':'):
        continue
    op = b.split()[0]
    k = classify(op)
    c[k] += 1
    if k in ('valu', 'salu'): ops[op] += 1
    if k == 'barrier':
        print("segment %2d: %5d  %s" % (seg, sum(c.values()), ", ".join("%s %d" % kv for kv in c.most_common())))
        print("      top:", ", ".join("%s %d" % kv for kv in ops.most_common(12)))
        c = collections.Counter(); ops = collections.Counter(); seg += 1
print("tail      : %5d  %s" % (sum(c.values()), ", ".join("%s %d" % kv for kv in c.most_common())))

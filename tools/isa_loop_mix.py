"""Dev tool: instruction mix of the loops of a kernel in an ISA listing (hipcc -S --cuda-device-only).
    python tools/isa_loop_mix.py file.s <substring of the mangled kernel name> ..."""
import re, sys, collections


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
for pat in sys.argv[2:]:
    for i, l in enumerate(lines):
        if l.startswith('_Z') and pat in l and l.rstrip().endswith(')') is False and ':' in l and '@' in l:
            start = i
            end = start + 1
            while not lines[end].startswith('.Lfunc_end'):
                end += 1
            body = lines[start:end]
            labels = {}
            for j, b in enumerate(body):
                m = re.match(r'^(\.LBB\d+_\d+):', b)
                if m:
                    labels[m.group(1)] = j
            loops = {}
            for j, b in enumerate(body):
                m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', b)
                if m and m.group(1) in labels and labels[m.group(1)] < j:
                    a = labels[m.group(1)]
                    loops[a] = max(loops.get(a, 0), j)
            print(l.split(':')[0][:110])
            for a, bnd in sorted(loops.items()):
                c = collections.Counter()
                for b in body[a:bnd + 1]:
                    b = b.strip()
                    if not b or b[0] in ';.':
                        continue
                    c[classify(b.split()[0])] += 1
                tot = sum(c.values())
                if tot > 60:
                    print("   loop of %5d instructions: %s" % (tot, ", ".join("%s %d" % kv for kv in c.most_common())))

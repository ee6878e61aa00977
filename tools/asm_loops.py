"""Instruction mix of every loop of one kernel in a hipcc -S listing: python tools/asm_loops.py file.s kernel_substring"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(L) if sys.argv[2] in l and re.match(r'^_Z\S+:', l)][0]
end = next(i for i in range(start, len(L)) if L[i].startswith('.Lfunc_end'))
body = L[start:end]
labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
print('kernel lines', len(body))
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i]
        c = collections.Counter(re.match(r'\s+([a-z0-9_]+)', x).group(1) for x in seg if re.match(r'\s+[a-z]', x))
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        print(m.group(1), 'instrs', sum(c.values()), 'valu', valu, 'ds', sum(v for k, v in c.items() if k.startswith('ds_')), 'waitcnt', c['s_waitcnt'])
        print('    ', c.most_common(14))

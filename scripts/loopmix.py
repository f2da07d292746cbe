"""Instruction mix of the MFMA loops of a kernel in a device assembly listing: python scripts/loopmix.py <file.s> <mangled-name substring>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for n in re.findall(r'^(_Z\S+):', s, re.M):
    if sys.argv[2] not in n:
        continue
    a = s.index(n + ':'); b = s.index('.Lfunc_end', a)
    body = s[a:b].splitlines()
    lab = {l[:-1].split()[0].rstrip(':'): i for i, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            loops.append((lab[m.group(1)], i))
    print(n[:90])
    for lo, hi in loops:
        c = Counter()
        for l in body[lo:hi + 1]:
            l = l.strip()
            if not l or l.startswith(('.', ';')) or l.endswith(':'):
                continue
            op = l.split()[0]
            k = 'mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else \
                'ds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_')) else op
            c[k] += 1
        if c['mfma']:
            print('   loop lines %d..%d:' % (lo, hi), dict(c))

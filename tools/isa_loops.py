"""Development aid: the loops of one kernel in the -save-temps ISA and their instruction mix.
    python tools/isa_loops.py <file.s> <mangled-name substring> [min instructions]
(build the .s with tools/isa_peek.sh or hipcc -save-temps)"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
nmin = int(sys.argv[3]) if len(sys.argv) > 3 else 100
L = open(path).read().split('\n')
start = next(k for k, l in enumerate(L) if re.match(r'^_ZN\S*' + re.escape(pat) + r'\S*:', l))
end = start
while '.end_amdhsa_kernel' not in L[end]:
    end += 1
body = L[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
for a, b in sorted(set(loops), key=lambda x: x[1] - x[0]):
    ops = collections.Counter()
    for l in body[a:b + 1]:
        t = l.strip()
        m = re.match(r'([a-z_0-9]+)\s', t + ' ')
        if m and not t.startswith('.') and not t.endswith(':') and not t.startswith(';'):
            ops[m.group(1)] += 1
    tot = sum(ops.values())
    if tot < nmin:
        continue
    cls = lambda f: sum(n for o, n in ops.items() if f(o))
    f64 = cls(lambda o: '_f64' in o)
    lane = cls(lambda o: 'readlane' in o or 'writelane' in o)
    smem = cls(lambda o: o.startswith('s_load') or o.startswith('s_buffer_load'))
    salu = cls(lambda o: o.startswith('s_')) - smem
    mem = cls(lambda o: o.startswith(('ds_', 'global_', 'scratch_', 'buffer_', 'flat_')))
    print("lines %5d-%5d  n=%4d  f64=%4d  lane=%3d  valu_other=%4d  salu=%4d  smem=%3d  ds/vmem=%3d  scratch=%d" %
          (a, b, tot, f64, lane, tot - f64 - lane - salu - smem - mem, salu, smem, mem,
           cls(lambda o: o.startswith('scratch_'))))

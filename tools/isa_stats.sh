#!/bin/bash
# Development aid: instruction mix of one kernel (mangled-name substring) from the ISA of a
# 12-band-only build:   tools/isa_stats.sh k_ffluxILi12ELb1ELb1 [-DFLAG=1 ...]
pat=${1:?kernel name substring}; shift
D=/tmp/isa_stats_$$; mkdir -p $D; cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -Wno-unused-value -save-temps -DBRUTUS_DEV_NB12_ONLY "$@" \
    /root/repo/brutus_amd/csrc/brutus_kernels.hip -o x.o 2>&1 | grep -E "error" | head
python3 - "$pat" "$D" <<'PY'
import re, sys, collections
pat, D = sys.argv[1], sys.argv[2]
L = open(D + '/brutus_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
start = next(k for k, l in enumerate(L) if re.match(r'^_ZN\S*' + re.escape(pat) + r'\S*:', l))
end = start
while '.end_amdhsa_kernel' not in L[end]:
    end += 1
ops = collections.Counter()
for l in L[start:end]:
    t = l.strip()
    m = re.match(r'([a-z_0-9]+)\s', t + ' ')
    if m and not t.startswith('.') and not t.endswith(':') and not t.startswith(';'):
        ops[m.group(1)] += 1
tot = sum(ops.values())
groups = collections.Counter()
for o, n in ops.items():
    g = ('f64' if re.search(r'_f64', o) else 'lane' if re.search(r'readlane|writelane|readfirstlane', o) else
         'cvt' if 'cvt' in o else 'ds' if o.startswith('ds_') else 'vmem' if o.startswith(('global_', 'buffer_', 'scratch_')) else
         'salu' if o.startswith('s_') else 'valu_other' if o.startswith('v_') else 'other')
    groups[g] += n
print(tot, "static instructions;", dict(groups))
print(ops.most_common(28))
for key in ['next_free_vgpr', 'next_free_sgpr', 'private_segment_fixed_size', 'group_segment_fixed_size']:
    for l in L[start:end + 60]:
        if key in l:
            print(l.strip()); break
PY
rm -rf $D

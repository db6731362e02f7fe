#!/bin/bash
# Development aid: compile the HIP translation unit with -save-temps and list the memory
# operations, waits, barriers and spills of one kernel (mangled-name substring).
#   tools/isa_peek.sh k_ffluxILi12ELb1ELb1
pat=${1:?kernel name substring}
D=/tmp/isa_peek; mkdir -p $D; cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -Wno-unused-value -save-temps \
    /root/repo/brutus_amd/csrc/brutus_kernels.hip -o x.o 2>&1 | grep -E "error" | head
python3 - "$pat" <<'PY'
import re, sys
pat = sys.argv[1]
L = open('/tmp/isa_peek/brutus_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
start = next(k for k, l in enumerate(L) if re.match(r'^_ZN\S*' + re.escape(pat) + r'\S*:', l))
end = start
while '.end_amdhsa_kernel' not in L[end]:
    end += 1
lines = L[start:end]
print(len(lines), "lines")
for k, l in enumerate(lines):
    t = l.strip()
    if re.search(r'global_load|s_barrier|s_waitcnt vmcnt|scratch_|global_store|buffer_|s_cbranch|^\.LBB', t):
        print(k, t[:100])
for key in ['next_free_vgpr', 'private_segment_fixed_size', 'group_segment_fixed_size']:
    for l in lines:
        if key in l:
            print(l.strip())
PY

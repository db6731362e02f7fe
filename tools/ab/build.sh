#!/bin/bash
# Build a variant of the library for same-box A/B timing (12-band instantiations only:
# seconds instead of a minute):   tools/ab/build.sh <name> [-DFLAG=1 ...]   -> tools/ab/<name>.so
# Run the variants with tools/ab/multi.sh on the GPU box.
name=${1:?variant name}; shift
R=$(cd $(dirname $0)/../.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value \
    -DBRUTUS_DEV_NB12_ONLY "$@" $R/brutus_amd/csrc/brutus_kernels.hip -o $R/tools/ab/$name.so

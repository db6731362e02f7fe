#!/bin/bash
# Development: recompile ONE translation unit of the library and relink (the other unit's object
# from the last full build is reused).   tools/relink_unit.sh pre32s [extra hipcc flags]
set -e
R=$(cd $(dirname $0)/.. && pwd); O=$R/build/libbrutus_amd.so.o; mkdir -p $O
case $1 in
  pre32s) src=$R/brutus_amd/csrc/pre32s_unit.hip; fl="-fno-slp-vectorize";;
  main) src=$R/brutus_amd/csrc/brutus_kernels.hip; fl="";;
esac
shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $fl "$@" -c $src -o $O/$(basename $src).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $O/brutus_kernels.hip.o $O/pre32s_unit.hip.o -o $R/brutus_amd/libbrutus_amd.so

#!/bin/bash
# Build a variant of the library for same-box A/B timing (12-band instantiations only:
# seconds instead of a minute):   tools/ab/build.sh <name> [-DFLAG=1 ...]   -> tools/ab/<name>.so
# Run the variants with tools/ab/multi.sh on the GPU box.
name=${1:?variant name}; shift
R=$(cd $(dirname $0)/../.. && pwd)
cd $R && python - "$name" "$@" <<'PY'
import sys
import __graft_entry__ as g
g.build_hip(force=True, out="tools/ab/%s.so" % sys.argv[1], extra=["-DBRUTUS_DEV_NB12_ONLY"] + sys.argv[2:])
PY

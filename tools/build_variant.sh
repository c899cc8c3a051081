#!/bin/bash
# tools/build_variant.sh <name> [-DFLAG=..]...  → opencl-raytracing_amd/variants/<name>.so
# (the four translation units of librt_amd.so with extra -D switches; __graft_entry__.build_hip does the work)
ROOT=$(dirname $(dirname $(readlink -f $0)))
NAME=$1; shift
mkdir -p $ROOT/opencl-raytracing_amd/variants
cd $ROOT && python3 - "$NAME" "$@" <<'PY'
import sys, os
import __graft_entry__ as g
name, extra = sys.argv[1], sys.argv[2:]
print(g.build_hip(force=True, extra=extra, out=os.path.join(g.PKG, "variants", name + ".so")))
PY

#!/bin/bash
# Build an alternative library for A/B runs on one box: the sources of git revision $1 for the
# files named after it, the working tree for everything else.
#   tools/ab_lib.sh HEAD conv_halo.hip  (EXTRA_FLAGS=-DX adds compiler flags; rev '-' = working tree only) ->  glsdet_amd/lib/ab/libglsdet_hip.so   (use: GLSDET_LIB_PATH=...)
set -e
rev=$1; shift
d=glsdet_amd/lib/ab; rm -rf $d; mkdir -p $d/csrc $d/include
cp glsdet_amd/csrc/*.hip glsdet_amd/csrc/*.h $d/csrc/
cp include/*.h $d/include/
if [ "$rev" != "-" ]; then for f in "$@"; do git show $rev:glsdet_amd/csrc/$f > $d/csrc/$f; done; fi
sed -i 's#"../../include/glsdet_hip.h"#"../include/glsdet_hip.h"#' $d/csrc/common.h
objs=""
for s in $d/csrc/*.hip; do o=${s%.hip}.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-gpu-rdc ${EXTRA_FLAGS:-} -c $s -o $o & objs="$objs $o"; done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libglsdet_hip.so $objs
rm -rf $d/csrc $d/include
ls -la $d

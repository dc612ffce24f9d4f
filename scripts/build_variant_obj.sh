#!/bin/bash
# Profiling build of the library with ONE other source recompiled: scripts/build_variant_obj.sh knrm NAME -DFLAG=... -> capreolus_amd/csrc/ablate/libcapreolus_amd_NAME.so
# (build_variant.sh is the bert.hip form; select with CAPAMD_LIB_PATH)
set -eu
src=$1; name=$2; shift 2
C=capreolus_amd/csrc
mkdir -p $C/ablate
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Iinclude -I$C "$@" $C/$src.hip -o $C/ablate/${src}_$name.o
objs=$(ls $C/*.o | grep -v "/$src.o" | grep -v "\.prof\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/ablate/libcapreolus_amd_$name.so $C/ablate/${src}_$name.o $objs
echo built $C/ablate/libcapreolus_amd_$name.so

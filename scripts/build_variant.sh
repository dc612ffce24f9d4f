#!/bin/bash
# Profiling build of the library: scripts/build_variant.sh NAME -DFLAG=... -> capreolus_amd/csrc/ablate/libcapreolus_amd_NAME.so
# (only bert.hip is recompiled; select with CAPAMD_LIB_PATH, see scripts/gemm_variants.sh)
set -eu
name=$1; shift
C=capreolus_amd/csrc
mkdir -p $C/ablate
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Iinclude -I$C "$@" $C/bert.hip -o $C/ablate/bert_$name.o
objs=$(ls $C/*.o | grep -v "/bert.o" | grep -v "\.prof\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/ablate/libcapreolus_amd_$name.so $C/ablate/bert_$name.o $objs
echo built $C/ablate/libcapreolus_amd_$name.so

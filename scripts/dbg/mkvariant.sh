#!/bin/bash
# scripts/dbg/mkvariant.sh <source.hip> <name> <-Dflags...>: a product-library variant under csrc/ablate/ (CAPAMD_LIB_PATH selects it)
set -e
C=/root/repo/capreolus_amd/csrc
src=$1; name=$2; shift 2
mkdir -p $C/ablate
base=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I$C "$@" -c $C/$src -o $C/ablate/${base}_$name.o
objs=$(ls $C/*.o | grep -v "\.prof\.o" | grep -v "/$base\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $C/ablate/libcapreolus_amd_$name.so $objs $C/ablate/${base}_$name.o
echo $C/ablate/libcapreolus_amd_$name.so

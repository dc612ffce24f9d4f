#!/bin/bash
# Registers / scratch of the kernels of one source: scripts/dbg/kernel_regs.sh SRC [pattern] [-D...]   (cross-compiles; no GPU needed)
set -u
src=$1; pat=${2:-.}; shift; shift || true
T=$(mktemp -d); C=capreolus_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Iinclude -I$C "$@" $C/$src.hip -o $T/x.o -save-temps=obj 2>&1 | grep -i "error" | head
S=$(ls $T/*gfx950.s)
for K in $(grep -o "^_Z[A-Za-z0-9_]*:" $S | tr -d ':' | grep -E "$pat"); do
  awk "/^$K:/,/\.end_amdhsa_kernel/" $S > $T/k.s
  grep -q amdhsa_next_free_vgpr $T/k.s || continue
  echo "$(echo $K | c++filt | cut -c1-110)  vgpr $(grep -o 'next_free_vgpr [0-9]*' $T/k.s | cut -d' ' -f2) scratch_bytes $(grep -o 'private_segment_fixed_size [0-9]*' $T/k.s | cut -d' ' -f2) scratch_ops $(grep -c scratch_ $T/k.s) lines $(wc -l < $T/k.s)"
done
[ -n "${KEEP_ASM:-}" ] && cp $S $KEEP_ASM
rm -rf $T

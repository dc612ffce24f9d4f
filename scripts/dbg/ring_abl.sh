#!/bin/bash
for v in "" ringabl1 ringabl2 ringabl3; do
  if [ -z "$v" ]; then unset CAPAMD_LIB_PATH; echo "== full"; else export CAPAMD_LIB_PATH=$PWD/capreolus_amd/csrc/ablate/libcapreolus_amd_$v.so; echo "== $v (1 = no DMA in loop, 2 = no fragment reads, 3 = neither)"; fi
  python scripts/dbg/ring_timeline.py 2>&1 | grep -v amdgpu | grep "ring\|pingpong" | grep "plain\|ffn2" | cut -c1-215
done

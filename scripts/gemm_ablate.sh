#!/bin/bash
# K-loop ablations of the ping-pong GEMM (profiling only): which of MFMA / LDS-DMA fill / LDS reads bounds a K step
for a in 0 1 2 4 3 5 6 7; do echo "ablate=$a (1 no-mfma, 2 no-fill, 4 no-reads)"; CAPAMD_GEMM_ABLATE=$a timeout 200 python scripts/gemm_timeline.py 2>&1 | grep -E "qkv-like|ffn2" | sed -E 's/.*deltas: \[([0-9]+), ([0-9]+), ([0-9]+), ([0-9]+).*total/\3 \4 total/'; done

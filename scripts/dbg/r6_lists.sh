#!/bin/bash
# round 6: list route after the work-list rebuild - parity tests, bench (KNRM, DRMM), kernel averages   (scripts/dbg/r6_lists.sh [spec ...])
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "lists or list_" 2>&1 | tail -5
[ $# -eq 0 ] && set -- base
bash scripts/dbg/lists_variants.sh "$@" 2>&1 | sed 's/void rocprim:: [0-9.]*  //g; s/void at::nativ [0-9.]*  //g; s/void at::nativ [0-9.]*$//'

#!/bin/bash
# lists route: parity tests, two bench rounds per model, kernel averages    (scripts/dbg/lists_quick.sh [name[:ENV=V,...][@libname] ...])
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "lists" 2>&1 | tail -3
[ $# -eq 0 ] && set -- base
bash scripts/dbg/lists_variants.sh "$@" 2>&1 | sed 's/void rocprim:: [0-9.]*  //g; s/void at::nativ [0-9.]*  //g; s/void at::nativ [0-9.]*$//'

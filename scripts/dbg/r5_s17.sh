#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --model bert --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-400
python -m pytest tests/test_gpu_bert.py -x -q 2>&1 | tail -2

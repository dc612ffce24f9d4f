#!/bin/bash
# BERT bench leg over the number of HIP streams the step's passages are split across, and the micro-batch size
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for cfg in "2 256" "3 256" "4 256" "2 128" "4 128" "1 256" "2 256"; do
  set -- $cfg
  timeout 600 python bench.py --steps 5 --warmup 2 --model bert --no-cpu-baseline --no-bert-other-dtype --bert-streams $1 --bert-microbatch $2 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json; r=json.load(open('/tmp/b.json')); print('streams $1 microbatch $2:', round(r['value'],1), 'docs/s', round(r['ms_per_step'],2), 'ms  executed', round(r['roofline']['whole_step_frac'],4), 'nominal', round(r['roofline']['whole_step_frac_nominal'],4))"
done 2>&1 | tee gpurun_out/bert_streams.txt

#!/bin/bash
# round 6: BERT bench leg over micro-batch sizes and stream counts (two alternating rounds on one box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for r in 1 2; do for spec in "256 2" "250 2" "200 2" "320 2" "400 2" "500 2" "128 2" "256 3" "500 3" "1000 2"; do set -- $spec
  v=$(timeout 600 python bench.py --steps 5 --warmup 2 --model bert --no-cpu-baseline --no-bert-other-dtype --bert-microbatch $1 --bert-streams $2 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.1f docs/s  %.2f ms  executed %.4f' % (r['value'], r['ms_per_step'], r['roofline']['whole_step_frac']))" 2>&1 | tail -1)
  echo "microbatch=$1 streams=$2  $v"
done; done 2>&1 | tee gpurun_out/bert_mb.txt

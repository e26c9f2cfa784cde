#!/bin/bash
# rollout-steps/s of small shards as 1..6 sub-batches on as many streams:  bash scripts/sweep_sub_batches.sh "256 512 1024" "1 2 3 4 6"
for g in $1; do for s in $2; do
  python bench.py --graphs $g --sub-batches $s --no-cpu-baseline --no-extras --steps 200 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graphs $g sub_batches $s:', round(d['value'],1), 'steps/s', round(d['ms_per_step'],4), 'ms')"
done; done

#!/bin/bash
# Kernel timeline of the small-shard rollout step (the 8-GPU strong-scaling shard is 256 graphs): durations AND gaps.
#   bash scripts/trace_small.sh <tag> [bench args]      (on the GPU box)
TAG=${1:-small}; shift
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 3 --preheat-s 0.2 --no-cpu-baseline --no-extras "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
f=$(ls $OUT/trace/*/*kernel_trace.csv | head -1)
python3 $R/scripts/trace_gaps.py $f > $OUT/gaps.txt
grep '^{' $OUT/trace.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("bench under trace:", d["value"], d["ms_per_step"])' >> $OUT/gaps.txt
rm -rf $OUT/trace
cat $OUT/gaps.txt

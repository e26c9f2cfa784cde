#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash scripts/profile_gpu.sh <tag> [bench args...]
# 1) rocprofv3 --kernel-trace --stats of the default bench command;
# 2) two separate --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) for HBM bytes.
# Everything lands under gpurun_out/<tag>/; scripts/summarize_profile.py turns it into profiles/<tag>_*.
set -o pipefail
TAG=${1:-r01}; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# stamp of what is being profiled (bench.py refuses to report traffic.json for other sources / workloads)
python3 - "$OUT" "$@" <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import bench
a = bench.parse(sys.argv[2:])
json.dump({'source_hash': bench.source_hash(), 'workload': f'{a.experiment}/{a.model}/{a.graphs}/n{a.neighbors}', 'split': int(not a.fp32_mfma)},
          open(os.path.join(sys.argv[1], 'stamp.json'), 'w'))
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --preheat-s 0.3 --no-cpu-baseline --no-extras "$@" > $OUT/trace.log 2>&1 || exit 1
grep '^{' $OUT/trace.log > $OUT/trace_bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --preheat-s 0 --no-cpu-baseline --no-extras "$@" > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --preheat-s 0 --no-cpu-baseline --no-extras "$@" > $OUT/pmc_write.log 2>&1 || exit 1
# keep only our kernels' rows of the (large) counter CSVs so they fit the gpurun_out merge limit
for d in pmc_fetch pmc_write; do
  f=$(ls $OUT/$d/*/*counter_collection.csv | head -1)
  head -1 $f > $OUT/$d.csv; grep -E 'msmp::' $f >> $OUT/$d.csv; rm -rf $OUT/$d
done
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv
rm -rf $OUT/trace
ls -la $OUT

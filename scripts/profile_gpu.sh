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
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/trace.log 2>&1 || exit 1
grep '^{' $OUT/trace.log > $OUT/trace_bench.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1 || exit 1
# keep only our kernels' rows of the (large) counter CSVs so they fit the gpurun_out merge limit
for d in pmc_fetch pmc_write; do
  f=$(ls $OUT/$d/*/*counter_collection.csv | head -1)
  head -1 $f > $OUT/$d.csv; grep -E 'msmp::' $f >> $OUT/$d.csv; rm -rf $OUT/$d
done
f=$(ls $OUT/trace/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv
rm -rf $OUT/trace
ls -la $OUT

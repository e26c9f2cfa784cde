#!/bin/bash
# Per-kernel A/B of two builds of the library under rocprofv3 --kernel-trace in ONE gpurun call:
#   bash scripts/ab_trace.sh <tag> libA.so libB.so [bench args]      -> gpurun_out/<tag>_<lib>.txt (top kernels: calls, avg us)
TAG=$1; A=$2; B=$3; shift 3
R=$PWD; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
for L in $A $B $A $B; do
  export MSMP_LIB_PATH=$R/msmp-pde_amd/$L
  D=/tmp/abtr_$$_$L; rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-extras "$@" > $D.log 2>&1 || { tail -5 $D.log; exit 1; }
  f=$(ls $D/*/*kernel_stats.csv | head -1)
  echo "== $L" >> $R/gpurun_out/${TAG}.txt
  python3 - "$f" >> $R/gpurun_out/${TAG}.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:5]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
  grep -o '"ms_per_step": [0-9.]*' $D.log | tail -1 >> $R/gpurun_out/${TAG}.txt
done
cat $R/gpurun_out/${TAG}.txt

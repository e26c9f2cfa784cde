#!/bin/bash
# bash scripts/duty_cycle.sh <tag>  -> gpurun_out/<tag>.txt : kernel averages at idle gaps of 0 / 1 / 3 x the step time
TAG=$1; R=$PWD; export TMPDIR=/tmp; cd /tmp
for G in 0 1 3; do
  D=/tmp/duty_$$_$G; rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/scripts/duty_cycle.py $G > $D.log 2>&1 || { tail -5 $D.log; exit 1; }
  f=$(ls $D/*/*kernel_stats.csv | head -1)
  echo "== idle gap of $G x the step time behind every step: $(grep gap $D.log)" >> $R/gpurun_out/$TAG.txt
  python3 - "$f" >> $R/gpurun_out/$TAG.txt <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    if 'msmp::' in r['Name'] or 'sleep' in r['Name'].lower() or 'spin' in r['Name'].lower():
        print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
cat $R/gpurun_out/$TAG.txt

#!/bin/bash
# kernel table of one MSMP-PDE training iteration (E2): bash scripts/trace_train512.sh <tag> [batch]  -> gpurun_out/<tag>.txt
TAG=$1; B=${2:-512}; R=$PWD; mkdir -p $R/gpurun_out; export TMPDIR=/tmp; cd /tmp
D=/tmp/trtr_$$; rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/scripts/train_profile_512.py $B > $D.log 2>&1 || { tail -5 $D.log; exit 1; }
f=$(ls $D/*/*kernel_stats.csv | head -1)
grep "ms per training" $D.log > $R/gpurun_out/$TAG.txt
python3 - "$f" >> $R/gpurun_out/$TAG.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows); n = 7.0
print(f'{sum(int(r["Calls"]) for r in rows) / n:.0f} launches, {tot / n / 1e6:.2f} ms of kernel time per iteration (7 iterations traced)')
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs']) / n / 1e3:9.1f} us/iter {int(r['Calls']) / n:7.1f} calls/iter {float(r['AverageNs']) / 1e3:8.1f} us/call  {r['Name'][:110]}")
PY
cat $R/gpurun_out/$TAG.txt

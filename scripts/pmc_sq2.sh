#!/bin/bash
# Second SQ counter pass (LDS / issue-stall breakdown): bash scripts/pmc_sq2.sh <tag> [bench args]
TAG=${1:-sq2}; shift
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
f=$(ls $OUT/pmc/*/*counter_collection.csv | head -1)
head -1 $f > $OUT/sq.csv; grep -E 'msmp::' $f >> $OUT/sq.csv; rm -rf $OUT/pmc
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open('$OUT/sq.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('msmp::', '')
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[k]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if 'SQ_WAVE_CYCLES' not in m: continue
    print(k[:40], ' '.join(f'{n}={v:.4g}' for n, v in m.items()))
PY

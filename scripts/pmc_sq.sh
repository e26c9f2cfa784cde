#!/bin/bash
# SQ counter pass for our kernels (run on the GPU box): bash scripts/pmc_sq.sh <tag> [bench args]
TAG=${1:-sq}; shift
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
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
    wc = m['SQ_WAVE_CYCLES']
    print(f"{k[:44]:44s} dur {m['dur_us']:8.1f} us  wave_cyc {wc:.3e}  wait_any {m['SQ_WAIT_ANY']/wc:5.2f}  wait_inst {m['SQ_WAIT_INST_ANY']/wc:5.2f}  active {m['SQ_ACTIVE_INST_ANY']/wc:5.2f}  mfma_busy {m['SQ_VALU_MFMA_BUSY_CYCLES']:.3e}  busy_cyc {m['SQ_BUSY_CYCLES']:.3e}  valu_insts {m['SQ_INSTS_VALU']:.3e}  lds_conf {m['SQ_LDS_BANK_CONFLICT']:.3e}")
PY

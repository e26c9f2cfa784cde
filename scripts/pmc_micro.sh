#!/bin/bash
# SQ counter passes over the isolated message kernels (scripts/edge_micro.py): bash scripts/pmc_micro.sh <tag>
# Two --pmc passes (8 SQ slots each); per kernel: busy / wait / issue fractions, MFMA-pipe share, mean VMEM and LDS latencies.
TAG=${1:-pmc_micro}; shift
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- python3 $R/scripts/edge_micro.py --reps 3 "$@" > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p2 -- python3 $R/scripts/edge_micro.py --reps 3 "$@" > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
for p in p1 p2; do f=$(ls $OUT/$p/*/*counter_collection.csv | head -1); head -1 $f > $OUT/$p.csv; grep -E 'msmp::(edge|node_proj)' $f >> $OUT/$p.csv; rm -rf $OUT/$p; done
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ('p1', 'p2'):
    for r in csv.DictReader(open('$OUT/' + p + '.csv')):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('msmp::', '') + ' grid ' + r.get('Grid_Size', '?')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[k]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    print(f"{k[:70]:70s} dur {m['dur_us']:7.1f} us  wait_any {m.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} "
          f"valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} lds {m.get('SQ_ACTIVE_INST_LDS',0)/wc:5.2f} mfma_busy/busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/max(m.get('SQ_BUSY_CYCLES',1),1)/4:5.2f}  "
          f"vmem_lat {m.get('SQ_INST_LEVEL_VMEM',0)/max(m.get('SQ_INSTS_VMEM',1),1):7.0f} lds_lat {m.get('SQ_INST_LEVEL_LDS',0)/max(m.get('SQ_INSTS_LDS',1),1):6.0f} "
          f"insts valu {m.get('SQ_INSTS_VALU',0):.3e} lds {m.get('SQ_INSTS_LDS',0):.3e} vmem {m.get('SQ_INSTS_VMEM',0):.3e} wait_lds {m.get('SQ_WAIT_INST_LDS',0)/wc:5.2f} conf {m.get('SQ_LDS_BANK_CONFLICT',0):.3e} busy {m.get('SQ_BUSY_CYCLES',0):.3e} wave_cyc {wc:.3e}")
PY

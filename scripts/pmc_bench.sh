#!/bin/bash
# SQ counter passes over the default bench step (all kernels of a rollout step): bash scripts/pmc_bench.sh <tag> [bench args]
# Per kernel: wave-cycle shares (waiting on memory / on an instruction, executing VALU / LDS), VALU and LDS instruction counts,
# mean VMEM latency.  SQ cycle counters tick in units of 4 clocks.
TAG=${1:-pmc_bench}; shift
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --preheat-s 0 --no-cpu-baseline --no-extras $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- $B > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p2 -- $B > $OUT/p2.log 2>&1 || { tail -5 $OUT/p2.log; exit 1; }
for p in p1 p2; do f=$(ls $OUT/$p/*/*counter_collection.csv | head -1); head -1 $f > $OUT/$p.csv; grep -E 'msmp::' $f >> $OUT/$p.csv; rm -rf $OUT/$p; done
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ('p1', 'p2'):
    for r in csv.DictReader(open('$OUT/' + p + '.csv')):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('msmp::', '') + ' grid ' + r.get('Grid_Size', '?')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[k]['dur_us'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        agg[k]['wg'].append(float(r.get('Workgroup_Size', 256)))
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1]['dur_us'])):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    print(f"{k[:64]:64s} x{len(c['dur_us'])//2:3d} dur {m['dur_us']:7.1f} us  wait_mem {m.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} "
          f"valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} lds {m.get('SQ_ACTIVE_INST_LDS',0)/wc:5.2f} mfma_busy_cyc {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0):.3e} busy {m.get('SQ_BUSY_CYCLES',0):.3e} wave_cyc {wc:.3e} "
          f"vmem_lat {m.get('SQ_INST_LEVEL_VMEM',0)/max(m.get('SQ_INSTS_VMEM',1),1):6.0f} insts valu {m.get('SQ_INSTS_VALU',0):.3e} salu {m.get('SQ_INSTS_SALU',0):.3e} lds {m.get('SQ_INSTS_LDS',0):.3e} vmem {m.get('SQ_INSTS_VMEM',0):.3e} conf {m.get('SQ_LDS_BANK_CONFLICT',0):.3e}")
PY

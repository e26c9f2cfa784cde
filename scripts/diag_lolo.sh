#!/bin/bash
# bash scripts/diag_lolo.sh  (on the GPU box): error profile + step time of the default build and the two lo*lo diagnostic builds
R=$PWD; O=$R/gpurun_out/diag_lolo; mkdir -p $O
for v in "" lolo1 lolo2; do
  lib=$R/msmp-pde_amd/libmsmp_pde${v:+_$v}.so
  [ -f $lib ] || { echo "missing $lib"; exit 1; }
  MSMP_LIB_PATH=$lib python3 scripts/diag_lolo.py > $O/err_${v:-default}.txt 2>&1 || { tail -5 $O/err_${v:-default}.txt; exit 1; }
  grep '^#' $O/err_${v:-default}.txt
  MSMP_LIB_PATH=$lib python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_${v:-default}.json 2> $O/bench_${v:-default}.err || { tail -5 $O/bench_${v:-default}.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$O/bench_${v:-default}.json').read().strip().splitlines()[-1]); print('bench ${v:-default}', d['value'], d['ms_per_step'])"
done

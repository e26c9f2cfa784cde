#!/bin/bash
# Kernel-level profiles (rocprofv3 trace + PMC traffic) of the BASELINE configs other than the default one, ON THE GPU BOX:
#   bash scripts/profile_configs.sh <tag-prefix>
# writes gpurun_out/<prefix>_{rpu,we3,we3_256,mswg3}/ (scripts/summarize_profile.py <tag> --no-traffic-json turns each into profiles/).
set -o pipefail
P=${1:-r04a}
bash scripts/profile_gpu.sh ${P}_rpu --experiment RPU --model MSMP-PDE2D > gpurun_out/${P}_rpu.log 2>&1 || exit 1
echo rpu done
bash scripts/profile_gpu.sh ${P}_we3 --experiment WE3 --model MSMP-PDE > gpurun_out/${P}_we3.log 2>&1 || exit 1
echo we3 done
bash scripts/profile_gpu.sh ${P}_we3_256 --experiment WE3 --model MSMP-PDE --graphs 256 > gpurun_out/${P}_we3_256.log 2>&1 || exit 1
echo we3_256 done
bash scripts/profile_gpu.sh ${P}_mswg3 --experiment MSWG3 --model MSMP-PDE2D > gpurun_out/${P}_mswg3.log 2>&1 || exit 1
echo mswg3 done

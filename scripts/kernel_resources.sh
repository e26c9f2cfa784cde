#!/bin/bash
# Register / spill / LDS table of the kernels of one source file:  scripts/kernel_resources.sh msmp-pde_amd/csrc/tile_kernels.hip [grep pattern] [extra flags]
src=$(realpath "$1"); pat=${2:-.}; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I "$root/include" -I "$root/msmp-pde_amd/csrc" "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o /tmp/kres.o 2>&1 \
  | grep "remark:" | sed 's/.*remark: *//; s/ \[-Rpass.*//' \
  | awk '/Function Name/{if(n)print n,v,a,sc,oc,ss,vs,l; n=$3} /^VGPRs:/{v="vgpr="$2} /^AGPRs:/{a="agpr="$2} /ScratchSize/{sc="scratch="$3} /Occupancy/{oc="occ="$3} /SGPRs Spill/{ss="sspill="$3} /VGPRs Spill/{vs="vspill="$3} /LDS Size/{l="lds="$4} END{print n,v,a,sc,oc,ss,vs,l}' \
  | grep -E "$pat"

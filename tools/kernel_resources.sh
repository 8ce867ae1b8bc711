#!/bin/bash
# VGPR / SGPR / scratch / occupancy of every kernel in one .hip file of the engine (cross-compiles, no GPU needed)
#   tools/kernel_resources.sh k_sor.hip [name-filter] [extra hipcc flags]
cd "$(dirname "$0")/../flowreg3d_amd/csrc" || exit 1
f=$1; filt=${2:-.}; shift; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-pass-failed "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/_kr.o 2>&1 |
  grep "Function Name\| VGPRs:\|AGPRs\|ScratchSize\|Occupancy\|LDS Size" |
  sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - - | grep "$filt" |
  sed 's/Function Name: //' | while read -r name rest; do echo "$(echo "$name" | c++filt | cut -c1-90) | $rest"; done

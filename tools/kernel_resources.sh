#!/bin/bash
# VGPRs / SGPRs / occupancy / LDS / scratch of every kernel in a .hip file, as hipcc reports them for gfx950
# (no GPU needed):   tools/kernel_resources.sh dtv-utils_amd/csrc/papr_sweep.hip [grep-pattern]
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=${1:?usage: kernel_resources.sh file.hip [pattern]}
PAT=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I"$R/include" -I"$R/dtv-utils_amd/csrc" \
    -c "$SRC" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" | sed 's/ \[-Rpass-analysis.*//' |
  awk '/Function Name:/ {name=$NF}
       /TotalSGPRs:/ {s=$NF} / VGPRs:/ {v=$NF} /AGPRs:/ {a=$NF} /ScratchSize/ {sc=$NF}
       /Occupancy/ {o=$NF} /LDS Size/ {l=$NF; printf "%s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %2s lds %6s\n", name, v, a, s, sc, o, l}' |
  while read -r sym rest; do
    printf "%-64s %s\n" "$(echo "$sym" | c++filt | sed 's/(.*//; s/^void //' | cut -c1-64)" "$rest"
  done | grep -E "$PAT"

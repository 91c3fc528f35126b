#!/bin/bash
# A/B two builds of libpaprhip.so on the SAME GPU box, interleaved (box-to-box and process-to-process
# spread on the one-sweep kernel is +-4 %, more than most code changes are worth):
#   cp dtv-utils_amd/libpaprhip.so dtv-utils_amd/libpaprhip_a.so      # build A
#   ... edit, make lib ...
#   cp dtv-utils_amd/libpaprhip.so dtv-utils_amd/libpaprhip_b.so      # build B
#   gpurun -- 'bash tools/ab_probe.sh "--variants 13 --bands 14"'
# (restores build B as libpaprhip.so at the end)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for round in 1 2 3; do
  for v in a b; do
    cp dtv-utils_amd/libpaprhip_$v.so dtv-utils_amd/libpaprhip.so
    echo "== build $v"
    timeout 300 python tools/sweep_probe.py --reps 10 ${1:---variants 13} 2>&1 | grep "v=\|two-pass" |
        sed "s/resolved=True counts_exact=True/OK/; s/aux.*wall/wall/" | cut -c1-130
  done
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s7; mkdir -p $O
timeout 900 python tools/sweep_probe.py --reps 10 --bands 14 \
  --configs 4:1024,20:512,20:1024,20:1280,21:1024,21:1536,22:512,22:768,23:512,23:768,23:1024,24:256,24:1024,25:512,25:768,26:1024,26:2048,27:1024,27:1536,28:256,28:1024,29:1024,4:1024 \
  > $O/sweep_probe.txt 2>&1; grep "v=\|two-pass" $O/sweep_probe.txt | cut -c1-75 ; grep -c "counts_exact=True trackers=True" $O/sweep_probe.txt; grep -c "v=" $O/sweep_probe.txt

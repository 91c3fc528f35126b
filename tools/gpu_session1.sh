#!/bin/bash
# round-2 GPU session 1: regression tests of the default paths, then geometry A/B of the sweep kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s1; mkdir -p $O
rocminfo | grep -m3 "Marketing\|gfx" > $O/box.txt; df -h . /tmp /dev/shm >> $O/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python tools/exact_sweep_probe.py --reps 10 > $O/exact_probe.txt 2>&1; tail -30 $O/exact_probe.txt
timeout 900 python tools/sweep_probe.py --reps 10 --bands 14 \
  --configs 4:1024,4:256,4:512,13:512,13:1024,32:256,33:256,34:256,35:256,36:256,36:512,37:512,38:256,39:256,40:256,41:256,42:512,43:1024,32:512,4:1024 \
  > $O/sweep_probe.txt 2>&1; grep "v=\|two-pass" $O/sweep_probe.txt | cut -c1-200

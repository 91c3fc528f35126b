#!/bin/bash
# GPU session 17: ballot-compaction stash (variants 36-39) — correctness, then timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q -k "geometries or constant or randomised or nan" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 900 python tools/sweep_probe.py --configs "4:1024,36:1024,37:1024,38:1024,39:512,4:1024,36:1024,37:1024,5:1024" --bands 14 --reps 10 > $O/sweep_probe.txt 2>&1; grep -v amdgpu.ids $O/sweep_probe.txt | cut -c1-170

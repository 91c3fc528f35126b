#!/bin/bash
# GPU session 12: loader / binner split sweep kernel — correctness, then timing against the default
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q -k "geometries or first_index or randomised" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 900 python tools/sweep_probe.py --configs "4:1024,71:256,71:512,72:256,72:512,72:1024,73:256,73:512,74:256,75:256,4:1024" --bands 14 --reps 10 > $O/sweep_probe.txt 2>&1; grep -v amdgpu.ids $O/sweep_probe.txt | cut -c1-170

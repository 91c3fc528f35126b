#!/bin/bash
# GPU session 14: 16-byte stash spills in papr_sweep_kernel (variants 5, 18, 19) — correctness, then timing
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 900 python tools/sweep_probe.py --configs "4:1024,5:1024,18:1024,19:512,4:1024,5:1024,18:1024,5:512" --bands 14 --reps 10 > $O/sweep_probe.txt 2>&1; grep -v amdgpu.ids $O/sweep_probe.txt | cut -c1-170

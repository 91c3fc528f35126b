#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s31; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q -k "hostile" > $O/pytest.log 2>&1; tail -30 $O/pytest.log | cut -c1-250

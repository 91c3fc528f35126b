#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s6; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "analyze or cli_tree_sum or one_sweep_equals or ts_ or scan" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
timeout 600 python tools/ablation_probe.py --reps 10 > $O/ablation.txt 2>&1; cat $O/ablation.txt | cut -c1-120

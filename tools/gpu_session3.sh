#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py -m gpu -q -k "arbitrary_level or torchrun_single or full_size_line" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 900 python tools/sweep_probe.py --reps 10 --bands 14 \
  --configs 4:1024,8:1024,8:256,6:512,32:256,33:256,34:256,35:256,40:256,41:256,36:512,37:512,42:512,43:1024,4:1024 \
  > $O/sweep_probe.txt 2>&1; grep "v=\|two-pass" $O/sweep_probe.txt | cut -c1-175
timeout 600 python tools/exact_sweep_probe.py --reps 10 --variants 48,49,50 > $O/exact_probe.txt 2>&1; grep -v "^n=.*counts_ok=True" $O/exact_probe.txt | cut -c1-250
bash tools/pmc_probe.sh s3/pmc "4:1024,32:256,41:256" graph > $O/pmc.log 2>&1; tail -5 $O/pmc.log

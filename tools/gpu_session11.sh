#!/bin/bash
# GPU session 11: O_DIRECT ingest through io_uring / reader threads (tests), cold-file ingest probe
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "o_direct or ingest" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 1500 bash tools/cold_ingest_probe.sh 10 > $O/cold_ingest.txt 2>&1; cat $O/cold_ingest.txt | cut -c1-600

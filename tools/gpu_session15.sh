#!/bin/bash
# GPU session 15: exact-sum one-sweep ingest of streamed shards (read-back of the few tiles the program needs again)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q -k "streamed or streams or ingest or cli" > $O/pytest.log 2>&1; tail -30 $O/pytest.log | cut -c1-300

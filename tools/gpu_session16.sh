#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s16; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 900 bash tools/cli_stream_probe.sh > $O/cli_stream.txt 2>&1; cat $O/cli_stream.txt | cut -c1-330

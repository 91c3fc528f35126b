#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s2; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $R/$O/counters.txt 2>&1)
grep -c . $O/counters.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log

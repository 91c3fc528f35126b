#!/bin/bash
# GPU session 38: timeline of one exact-sum step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s38; mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python bench.py --exact --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $O/prof.json 2> $O/prof.err < /dev/null
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
if [ -n "$T" ]; then python tools/step_timeline.py "$T" | tee $O/timeline_exact.txt; else echo "no kernel trace"; tail -5 $O/prof.err; fi
rm -rf $O/prof

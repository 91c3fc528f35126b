#!/bin/bash
# GPU session 34: whole-chip recount geometry (2 x 1024 threads per CU) + the timeline of a step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s34; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
for ROUND in 1 2; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json"))
    print("round $ROUND: step %.4f / %.4f  kernel %.4f / %.4f  aux %.4f / %.4f host %.4f / %.4f parity %s %s" % (d["ms_per_step"], d["graph"]["ms_per_step"], d["roofline"]["kernel_ms"], d["graph"]["roofline"]["kernel_ms"], d["kernels"]["estimate_and_recount_kernels"]["ms_per_step"], d["graph"]["kernels"]["estimate_and_recount_kernels"]["ms_per_step"], d["kernels"]["host_and_exchange_ms_per_step"], d["graph"]["kernels"]["host_and_exchange_ms_per_step"], d["parity_in_run"], d["graph"]["parity_in_run"]))
except Exception as e:
    print("failed", e, open("$O/b.err").read()[-800:])
PY
done
export TMPDIR=/tmp
rm -rf $O/prof; rocprofv3 --kernel-trace --stats -d $O/prof -o t -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $O/prof.json 2> $O/prof.err
T=$(find $O/prof -name '*kernel_trace.csv' | head -1); S=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python tools/step_timeline.py $T | tee $O/timeline.txt
cut -c1-60,200- $S | head -12
cp $S $O/kernel_stats.csv; rm -rf $O/prof

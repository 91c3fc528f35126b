#!/bin/bash
# GPU session 39: shuffle scans in the exact-sum helper kernels (spec scan, plan)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s39; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; tail -3 $O/pytest.log | cut -c1-300
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d["graph"]
    print("%-24s step %.4f / %.4f  kernel %.4f / %.4f  aux %.4f / %.4f host %.4f / %.4f parity %s %s redo %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["kernels"]["estimate_and_recount_kernels"]["ms_per_step"], g["kernels"]["estimate_and_recount_kernels"]["ms_per_step"],
        d["kernels"]["host_and_exchange_ms_per_step"], g["kernels"]["host_and_exchange_ms_per_step"],
        d["parity_in_run"], g["parity_in_run"], d["config"]["one_sweep"]["exact_redo_tiles_per_step"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2 3; do
  timeout 300 python bench.py --exact --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/bx$ROUND.json 2> $O/bx.err < /dev/null
  line "r$ROUND exact" $O/bx$ROUND.json
done
export TMPDIR=/tmp
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python bench.py --exact --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $O/prof.json 2> $O/prof.err < /dev/null
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
if [ -n "$T" ]; then python tools/step_timeline.py "$T" | head -21 | tee $O/timeline_exact.txt; else echo "no kernel trace"; tail -5 $O/prof.err; fi
rm -rf $O/prof

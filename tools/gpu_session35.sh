#!/bin/bash
# GPU session 35: kernel timing bound to the dispatch (hipExtLaunchKernelGGL) instead of event records, the sweep's
# counters and the recount's histogram handed to the host by the kernels themselves; A/B against the old forms
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s35; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; tail -3 $O/pytest.log | cut -c1-300
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d["graph"]
    print("%-28s step %.4f / %.4f  kernel %.4f / %.4f  aux %.4f / %.4f host %.4f / %.4f parity %s %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["kernels"]["estimate_and_recount_kernels"]["ms_per_step"], g["kernels"]["estimate_and_recount_kernels"]["ms_per_step"],
        d["kernels"]["host_and_exchange_ms_per_step"], g["kernels"]["host_and_exchange_ms_per_step"],
        d["parity_in_run"], g["parity_in_run"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2 3; do
  for CFG in "1 0" "0 0" "1 1" "0 1"; do
    set -- $CFG
    PAPR_EXT_TIMING=$1 PAPR_FUSED_COPIES=$2 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err < /dev/null
    line "r$ROUND ext=$1 copies=$2" $O/b.json
  done
done
PAPR_EXT_TIMING=1 timeout 300 python bench.py --exact --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/bx.json 2> $O/bx.err < /dev/null
line "exact ext=1" $O/bx.json
PAPR_EXT_TIMING=0 PAPR_FUSED_COPIES=1 timeout 300 python bench.py --exact --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/bx.json 2> $O/bx.err < /dev/null
line "exact ext=0 copies=1" $O/bx.json
export TMPDIR=/tmp
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $O/prof.json 2> $O/prof.err < /dev/null
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
if [ -n "$T" ]; then python tools/step_timeline.py "$T" | tee $O/timeline.txt; else echo "no kernel trace"; tail -5 $O/prof.err; fi
rm -rf $O/prof

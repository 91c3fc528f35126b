#!/bin/bash
# GPU session 36: what the timing brackets themselves cost a step (PAPR_TIME_KINDS), and the finalize kernel's copy of
# the sweep's counters against the D2H copy (PAPR_FUSED_COPIES)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s36; mkdir -p $O
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d["graph"]
    print("%-34s step %.4f / %.4f  kernel %.4f / %.4f  aux %.4f / %.4f host %.4f / %.4f parity %s %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["kernels"]["estimate_and_recount_kernels"]["ms_per_step"], g["kernels"]["estimate_and_recount_kernels"]["ms_per_step"],
        d["kernels"]["host_and_exchange_ms_per_step"], g["kernels"]["host_and_exchange_ms_per_step"],
        d["parity_in_run"], g["parity_in_run"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2 3; do
  for CFG in "255 0" "255 1" "8 0" "0 0"; do
    set -- $CFG
    PAPR_TIME_KINDS=$1 PAPR_FUSED_COPIES=$2 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err < /dev/null
    line "r$ROUND kinds=$1 stream_copies=$2" $O/b.json
  done
done
for K in 255 8; do
PAPR_TIME_KINDS=$K timeout 300 python bench.py --exact --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/bx.json 2> $O/bx.err < /dev/null
line "exact kinds=$K" $O/bx.json
done

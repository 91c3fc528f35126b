#!/bin/bash
# GPU session 40: exact-sum bench lines again (host-side noise check), default lines beside them
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s40; mkdir -p $O
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d["graph"]
    print("%-24s step %.4f / %.4f  kernel %.4f / %.4f  aux %.4f / %.4f host %.4f / %.4f parity %s %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["kernels"]["estimate_and_recount_kernels"]["ms_per_step"], g["kernels"]["estimate_and_recount_kernels"]["ms_per_step"],
        d["kernels"]["host_and_exchange_ms_per_step"], g["kernels"]["host_and_exchange_ms_per_step"],
        d["parity_in_run"], g["parity_in_run"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
nproc; uptime
for ROUND in 1 2 3 4; do
  timeout 300 python bench.py --exact --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/bx$ROUND.json 2> $O/bx.err < /dev/null
  line "r$ROUND exact" $O/bx$ROUND.json
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b$ROUND.json 2> $O/b.err < /dev/null
  line "r$ROUND default" $O/b$ROUND.json
done
uptime

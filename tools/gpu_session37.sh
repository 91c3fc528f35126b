#!/bin/bash
# GPU session 37: full suite + bench lines after the step-time work (recount geometry, finalize copy, timing level 2)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s37; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; tail -3 $O/pytest.log | cut -c1-300
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d["graph"]
    print("%-24s step %.4f / %.4f  kernel %.4f / %.4f  aux %.4f / %.4f host %.4f / %.4f parity %s %s frac %.4f" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["kernels"]["estimate_and_recount_kernels"]["ms_per_step"], g["kernels"]["estimate_and_recount_kernels"]["ms_per_step"],
        d["kernels"]["host_and_exchange_ms_per_step"], g["kernels"]["host_and_exchange_ms_per_step"],
        d["parity_in_run"], g["parity_in_run"], d["roofline"]["frac"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b$ROUND.json 2> $O/b.err < /dev/null
  line "r$ROUND default" $O/b$ROUND.json
  timeout 300 python bench.py --exact --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/bx$ROUND.json 2> $O/bx.err < /dev/null
  line "r$ROUND exact" $O/bx$ROUND.json
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.json 2> $O/driver.err < /dev/null
line "driver line" $O/driver.json

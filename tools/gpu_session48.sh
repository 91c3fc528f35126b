#!/bin/bash
# GPU session 48: exact-sum kernel with the explicit packed squares; exact tests + bench lines (default beside them)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s48; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py -m gpu -x -q -k "exact" > $O/pytest.log 2>&1 < /dev/null; tail -2 $O/pytest.log | cut -c1-200
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d.get("graph") or d
    print("%-18s step %.4f / %.4f  kernel %.4f / %.4f  frac %.4f / %.4f parity %s %s sum %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"], d["roofline"]["frac"], g["roofline"]["frac"],
        d["parity_in_run"], g["parity_in_run"], d["config"]["sum_hex"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2 3; do
  timeout 300 python bench.py --exact --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > $O/bx.json 2> $O/bx.err < /dev/null
  line "r$ROUND exact" $O/bx.json
done
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err < /dev/null
line "default" $O/b.json

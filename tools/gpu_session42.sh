#!/bin/bash
# GPU session 42: the new default sweep kernel (variant 40: 512 x 8, one workgroup per CU, branch-free ballot stash)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s42; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; tail -4 $O/pytest.log | cut -c1-300
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d.get("graph") or d
    print("%-24s step %.4f / %.4f  kernel %.4f / %.4f  frac %.4f / %.4f  host %.4f parity %s %s resolved %s gave_up %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["roofline"]["frac"], g["roofline"]["frac"], d["kernels"]["host_and_exchange_ms_per_step"],
        d["parity_in_run"], g["parity_in_run"], d["config"]["one_sweep"]["steps_resolved_from_the_sweep"], d["config"]["one_sweep"]["gave_up"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2 3; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b$ROUND.json 2> $O/b.err < /dev/null
  line "r$ROUND default" $O/b$ROUND.json
  PAPR_HIP_TUNE=wvariant=4 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b4_$ROUND.json 2> $O/b.err < /dev/null
  line "r$ROUND variant 4 (host path)" $O/b4_$ROUND.json
done
for SIG in bursty constant; do
  timeout 300 python bench.py --signal $SIG --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $O/b_$SIG.json 2> $O/b.err < /dev/null
  line "$SIG" $O/b_$SIG.json
done
timeout 300 python bench.py --force-miss --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $O/b_miss.json 2> $O/b.err < /dev/null
line "miss" $O/b_miss.json

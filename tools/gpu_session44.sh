#!/bin/bash
# GPU session 44: the exact-sum sweep kernel without branches in the per-sample code (variant 31) against variant 56
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s44; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q -k "exact_sweep_variants or exact_one_sweep or exact" > $O/pytest.log 2>&1 < /dev/null; tail -3 $O/pytest.log | cut -c1-300
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d.get("graph") or d
    print("%-24s step %.4f / %.4f  kernel %.4f / %.4f  frac %.4f / %.4f  parity %s %s redo %s sum %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["roofline"]["frac"], g["roofline"]["frac"], d["parity_in_run"], g["parity_in_run"],
        d["config"]["one_sweep"]["exact_redo_tiles_per_step"], d["config"]["sum_hex"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2 3; do
  for V in 56 31; do
    PAPR_HIP_TUNE=wvariant=$V timeout 300 python bench.py --exact --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/bx.json 2> $O/bx.err < /dev/null
    line "r$ROUND exact variant $V" $O/bx.json
  done
done

#!/bin/bash
# GPU session 32: the single-wait step in exact-sum mode (PAPR_FUSED_EXACT)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s32; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log | cut -c1-300
for ROUND in 1 2 3; do
for F in 1 0; do
  PAPR_FUSED_EXACT=$F timeout 300 python bench.py --exact --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json"))
    print("round $ROUND fused_exact=$F: step %.4f / %.4f  kernel %.4f / %.4f  host %.4f / %.4f parity %s %s redo %s sum %s" % (d["ms_per_step"], d["graph"]["ms_per_step"], d["roofline"]["kernel_ms"], d["graph"]["roofline"]["kernel_ms"], d["kernels"]["host_and_exchange_ms_per_step"], d["graph"]["kernels"]["host_and_exchange_ms_per_step"], d["parity_in_run"], d["graph"]["parity_in_run"], d["config"]["one_sweep"]["exact_redo_tiles_per_step"], d["config"]["sum_hex"]))
except Exception as e:
    print("failed", e, open("$O/b.err").read()[-800:])
PY
done
done

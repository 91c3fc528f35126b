#!/bin/bash
# GPU session 27: the fused estimate -> device guess -> sweep sequence (default step, no peers)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s27; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py tests/test_bench_contract.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
for ROUND in 1 2 3; do
for F in 1 0; do
  PAPR_FUSED_GUESS=$F timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json"))
    print("round $ROUND fused=$F: step %.4f / %.4f  kernel %.4f / %.4f  host %.4f / %.4f parity %s %s band %s resolved %s" % (d["ms_per_step"], d["graph"]["ms_per_step"], d["roofline"]["kernel_ms"], d["graph"]["roofline"]["kernel_ms"], d["kernels"]["host_and_exchange_ms_per_step"], d["graph"]["kernels"]["host_and_exchange_ms_per_step"], d["parity_in_run"], d["graph"]["parity_in_run"], d["config"]["one_sweep"]["band_log2"], d["config"]["one_sweep"]["steps_resolved_from_the_sweep"]))
except Exception as e:
    print("failed", e, open("$O/b.err").read()[-800:])
PY
done
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -25 $O/pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/s5/bench.json"))
print("default:", round(d["ms_per_step"],4), round(d["roofline"]["frac"],4), d["parity_in_run"], d["kernels"]["host_and_exchange_ms_per_step"], "graph:", round(d["graph"]["ms_per_step"],4), round(d["graph"]["roofline"]["frac"],4), d["graph"]["parity_in_run"])
PY
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --exact > $O/bench_exact.json 2> $O/bench_exact.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/s5/bench_exact.json"))
print("exact:", round(d["ms_per_step"],4), d["config"]["sum_hex"], d["parity_in_run"], d["kernels"]["host_and_exchange_ms_per_step"], "graph:", round(d["graph"]["ms_per_step"],4), d["graph"]["parity_in_run"])
PY

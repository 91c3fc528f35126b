#!/bin/bash
# GPU session 46: final check — full suite, smoke, the driver's bench command
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s46; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; tail -3 $O/pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as ge; ge.smoke(); print('smoke ok')" 2>&1 | tail -2
S=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.json 2> $O/driver.err < /dev/null; echo rc=$? wall=$(( $(date +%s) - S ))s lines=$(wc -l < $O/driver.json)
python - $O/driver.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); g = d["graph"]
print("step %.4f / %.4f  kernel %.4f / %.4f  frac %.4f / %.4f  parity %s %s  value %.0f  cpu %.1f  e2e %.3f s" % (d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"], d["roofline"]["frac"], g["roofline"]["frac"], d["parity_in_run"], g["parity_in_run"], d["value"], d["cpu_baseline"]["value"], d["e2e"]["default"]["seconds"]))
print(d["roofline"]["kernel"], d["config"]["workload"])
PY

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s8; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as ge; ge.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 2400 bash tools/collect_profiles_r02.sh r02 > $O/collect.log 2>&1; tail -3 $O/collect.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02/bench_*.json")):
    try:
        d=json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    g=d.get("graph",{})
    print(f.split("/")[-1], round(d["ms_per_step"],4), round(d["roofline"]["frac"],4), d.get("parity_in_run"), "| graph", round(g.get("ms_per_step",0),4), g.get("parity_in_run"), "| e2e" , {k:(round(v["seconds"],3), v["stdout_identical_to_reference"]) for k,v in d.get("e2e",{}).items() if isinstance(v,dict)})
PY

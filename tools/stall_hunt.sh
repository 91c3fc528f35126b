#!/bin/bash
# Run ON THE GPU BOX (through gpurun): how often does a step of the timed region stall (a step several times its median: the
# host's wait returning late, the GPU idling and clocking down behind it)?  N fresh processes per setting of the runtime's wait
# policy — interrupts (the default) and HSA_ENABLE_INTERRUPT=0 (the runtime polls its completion signals) — `bench.py
# --headline-only --steps 200`; one line per run and a tally.
#   gpurun -- 'bash tools/stall_hunt.sh 20'
set -u
N=${1:-20}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for MODE in default poll; do
  for i in $(seq 1 "$N"); do
    if [ "$MODE" = poll ]; then export HSA_ENABLE_INTERRUPT=0; else export HSA_ENABLE_INTERRUPT=1; fi
    python3 bench.py --headline-only --no-cpu-baseline --no-e2e --steps 200 --warmup 5 --full-json /tmp/stall_$MODE.json > /dev/null 2>&1
    python3 - "$MODE" "$i" /tmp/stall_$MODE.json <<'PY'
import json, sys
mode, i, path = sys.argv[1:4]
d = json.load(open(path))
out = []
for tag, r in (("default", d["roofline"]), ("graph", d["graph"]["roofline"])):
    s = r["step_ms"]
    out.append("%s: median %.3f max %.3f mean %.4f" % (tag, s["median"], s["max"], d["ms_per_step"] if tag == "default" else d["graph"]["ms_per_step"]))
print(mode, i, " | ".join(out), flush=True)
PY
  done
done

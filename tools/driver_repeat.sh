#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the driver's own command, verbatim, un-profiled, in fresh processes — the FIRST of
# them is the first GPU process of a fresh lease, which is what the driver's BENCH_rNN.json records.  Appends one line
# per run (lease tag, run number, wall seconds, the JSON) to gpurun_out/driver_repeat/<tag>.txt.
#   gpurun -- 'bash tools/driver_repeat.sh <tag> [runs]'
set -u
TAG=${1:-lease}
RUNS=${2:-3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/driver_repeat
mkdir -p "$O"
cd "$R"
for i in $(seq 1 "$RUNS"); do
  T0=$(date +%s.%N)
  LINE=$(python3 bench.py --gpus 1 --steps 20 --warmup 5 2> "$O/$TAG.run$i.err")
  T1=$(date +%s.%N)
  printf '{"lease": "%s", "run": %d, "first_gpu_process_of_the_lease": %s, "wall_s": %.1f, "line": %s}\n' \
      "$TAG" "$i" "$([ "$i" = 1 ] && echo true || echo false)" "$(python3 -c "print($T1 - $T0)")" "${LINE:-null}" >> "$O/$TAG.txt"
done
python3 - "$O/$TAG.txt" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l)
    b = j["line"]
    if not b:
        print(j["lease"], j["run"], "NO LINE"); continue
    r = b["roofline"]
    print(j["lease"], j["run"], "wall %.0fs" % j["wall_s"], "ms/step %.4f" % b["ms_per_step"], "value %.0f" % b["value"],
          "kernel_ms %.4f" % r["kernel_ms"], "frac %.4f" % r["frac"], "step_ms", r.get("step_ms"),
          "host %.4f" % r.get("host_and_exchange_ms_per_step", -1))
    for k, v in (r.get("legs") or {}).items():
        print("     ", k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "kernel_ms", "frac", "step_ms", "seconds", "ingest_frac_of_h2d_ceiling", "all_kernels_frac_of_peak")})
PY

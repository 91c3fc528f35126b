#!/bin/bash
# interleaved A/B of two library builds on one box: sweep kernel time (HIP events), default and -g, plain and --exact
#   gpurun -- 'bash tools/gpu_ab.sh build_bisect/lib_old.so ""'      ("" = the in-tree library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/ab; mkdir -p $O
A=${1:-}; B=${2:-}
for ROUND in 1 2 3; do
  for L in "$A" "$B"; do
    N=$(basename "${L:-intree}")
    for FLAGS in "" "--exact"; do
      PAPR_LIB_PATH=${L:+$R/$L} timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-e2e $FLAGS > $O/ab.json 2> $O/ab.err
      python - <<PY
import json
try:
    d=json.load(open("$O/ab.json"))
    print("round $ROUND %-12s %-8s kernel %.4f / -g %.4f   step %.4f / %.4f  parity %s %s" % ("$N", "$FLAGS" or "plain", d["roofline"]["kernel_ms"], d["graph"]["roofline"]["kernel_ms"], d["ms_per_step"], d["graph"]["ms_per_step"], d["parity_in_run"], d["graph"]["parity_in_run"]))
except Exception as e:
    print("$N", "$FLAGS", "failed", e)
PY
    done
  done
done

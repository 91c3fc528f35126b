#!/bin/bash
# interleaved A/B of PAPR_HIP_TUNE settings through bench.py (step time, kernel time, host share):
#   gpurun -- 'bash tools/gpu_ab_tune.sh "" "wvariant=24" [extra bench flags]'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/abtune; mkdir -p $O
A=$1; B=$2; shift 2
for ROUND in 1 2 3 4; do
for T in "$A" "$B"; do
  PAPR_HIP_TUNE="$T" timeout 300 python bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json"))
    print("round $ROUND tune '%s': step %.4f / %.4f  kernel %.4f / %.4f  host %.4f / %.4f parity %s %s" % ("$T", d["ms_per_step"], d["graph"]["ms_per_step"], d["roofline"]["kernel_ms"], d["graph"]["roofline"]["kernel_ms"], d["kernels"]["host_and_exchange_ms_per_step"], d["graph"]["kernels"]["host_and_exchange_ms_per_step"], d["parity_in_run"], d["graph"]["parity_in_run"]))
except Exception as e:
    print("failed", e, open("$O/b.err").read()[-600:])
PY
done
done

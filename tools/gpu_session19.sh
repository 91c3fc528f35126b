#!/bin/bash
# GPU session 19: exact sweep variants out of the MEASURE build (ballot ring +/- lean sum +/- wide batches)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s19; mkdir -p $O
export PAPR_LIB_PATH=$R/build_measure/libpaprhip_measure.so
for ROUND in 1 2; do
for V in 56 43 33 56 43 33; do
  PAPR_HIP_TUNE="wvariant=$V" timeout 300 python bench.py --exact --steps 30 --warmup 3 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json"))
    print("round $ROUND exact variant $V: kernel %.4f / -g %.4f  step %.4f / %.4f parity %s %s" % (d["roofline"]["kernel_ms"], d["graph"]["roofline"]["kernel_ms"], d["ms_per_step"], d["graph"]["ms_per_step"], d["parity_in_run"], d["graph"]["parity_in_run"]))
except Exception as e:
    print("$V failed", e, open("$O/b.err").read()[-600:])
PY
done
done

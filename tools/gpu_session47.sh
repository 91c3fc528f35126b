#!/bin/bash
# GPU session 47: the exact-sum kernel with the histogram and / or the stash branch-free (MEASURE variants 29 / 28 / 31) against 56
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s47; mkdir -p $O
export PAPR_LIB_PATH=$R/build_measure/libpaprhip.so
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d.get("graph") or d
    print("%-24s step %.4f / %.4f  kernel %.4f / %.4f  parity %s %s sum %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["parity_in_run"], g["parity_in_run"], d["config"]["sum_hex"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2; do
  for V in 56 29 28 31; do
    PAPR_HIP_TUNE=wvariant=$V timeout 300 python bench.py --exact --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > $O/bx.json 2> $O/bx.err < /dev/null
    line "r$ROUND exact variant $V" $O/bx.json
  done
done

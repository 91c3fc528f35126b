#!/bin/bash
# GPU session 43: variant 40 through the single-wait step (device-built table) and through the host path, against variant 4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s43; mkdir -p $O
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d.get("graph") or d
    print("%-28s step %.4f / %.4f  kernel %.4f / %.4f  frac %.4f / %.4f  host %.4f / %.4f parity %s %s stash %s / %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["roofline"]["frac"], g["roofline"]["frac"], d["kernels"]["host_and_exchange_ms_per_step"], g["kernels"]["host_and_exchange_ms_per_step"],
        d["parity_in_run"], g["parity_in_run"], d["config"]["one_sweep"]["stash_samples"], g["config"]["one_sweep"]["stash_samples"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2 3; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err < /dev/null
  line "r$ROUND v40 single-wait" $O/b.json
  PAPR_FUSED_GUESS=0 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err < /dev/null
  line "r$ROUND v40 host path" $O/b.json
  PAPR_HIP_TUNE=wvariant=4 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err < /dev/null
  line "r$ROUND v4 host path" $O/b.json
done

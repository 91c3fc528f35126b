#!/bin/bash
# GPU session 18: ballot ring stash in the sweep2 kernels (55: exact, 57: plain)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q -k "geometries or exact_sweep_variants or analyze" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for ROUND in 1 2; do
for V in 48 55; do
  PAPR_HIP_TUNE="wvariant=$V" timeout 300 python bench.py --exact --steps 30 --warmup 3 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json"))
    print("round $ROUND exact variant $V: kernel %.4f / -g %.4f  step %.4f / %.4f parity %s %s" % (d["roofline"]["kernel_ms"], d["graph"]["roofline"]["kernel_ms"], d["ms_per_step"], d["graph"]["ms_per_step"], d["parity_in_run"], d["graph"]["parity_in_run"]))
except Exception as e:
    print("$V failed", e, open("$O/b.err").read()[-800:])
PY
done
done
timeout 600 python tools/sweep_probe.py --configs "4:1024,32:256,57:256,4:1024" --bands 14 --reps 10 > $O/sweep_probe.txt 2>&1; grep -v amdgpu.ids $O/sweep_probe.txt | cut -c1-150

#!/bin/bash
# GPU session 45: the packet scan's forms — packets per lane x workgroup size x per-(wave, PID) aggregated table update
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s45; mkdir -p $O
for CFG in "1 1024 0" "1 1024 1" "2 1024 0" "2 1024 1" "4 1024 0" "4 1024 1" "2 512 0" "2 512 1" "4 512 0" "4 512 1" "8 512 0" "8 512 1" "1 1024 0" "4 512 1"; do
  set -- $CFG
  TS_SCAN_UNROLL=$1 TS_SCAN_BLOCK=$2 TS_SCAN_AGG=$3 timeout 300 python bench.py --workload ts --steps 30 --warmup 3 --no-cpu-baseline > $O/t.json 2> $O/t.err < /dev/null
  python - "$CFG" $O/t.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("unroll/block/agg %-10s step %.4f ms  kernel %.4f ms  frac %.4f  merge %.4f  %.0f Mpackets/s" % (sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"].get("merge_kernel_ms", 0), d["value"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
TS_SCAN_UNROLL=4 TS_SCAN_BLOCK=512 TS_SCAN_AGG=1 timeout 900 python -m pytest tests/test_ts_scan.py -m gpu -x -q 2>&1 | tail -2
TS_SCAN_UNROLL=1 TS_SCAN_BLOCK=1024 TS_SCAN_AGG=1 timeout 900 python -m pytest tests/test_ts_scan.py -m gpu -x -q 2>&1 | tail -2

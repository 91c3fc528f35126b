#!/bin/bash
# GPU session 9: give-up rule of the sweep (constant envelope), parallel TS merge, TS scan unroll, TS request counters
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/s9
rm -rf "$O"; mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python $R/bench.py --no-cpu-baseline --no-e2e"
for SIG in constant bursty; do
  timeout 300 $B --signal $SIG > $O/bench_$SIG.json 2> $O/bench_$SIG.err
  timeout 300 $B --signal $SIG --exact > $O/bench_${SIG}_exact.json 2> $O/bench_${SIG}_exact.err
done
timeout 300 $B > $O/bench_default.json 2> $O/bench_default.err
timeout 300 $B --exact > $O/bench_exact.json 2> $O/bench_exact.err
for U in 1 2 4; do
  TS_SCAN_UNROLL=$U timeout 300 $B --workload ts > $O/bench_ts_u$U.json 2> $O/bench_ts_u$U.err
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/pmc_tcc_ts -- \
    $B --workload ts --steps 2 --warmup 1 > $O/pmc_tcc_ts.json 2> $O/pmc_tcc_ts.err
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_hit_ts -- \
    $B --workload ts --steps 2 --warmup 1 > $O/pmc_hit_ts.json 2> $O/pmc_hit_ts.err
find $O -name "*.csv" -size +8M -delete
cd "$R"
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/s9/bench_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    g = d.get("graph") or {}
    print(os.path.basename(f), round(d["ms_per_step"], 4), round(d["roofline"]["frac"], 4), d["roofline"].get("kernel_ms"),
          d.get("parity_in_run"), "| graph", round(g.get("ms_per_step", 0), 4), (g.get("roofline") or {}).get("kernel_ms"),
          "|", (d["config"].get("one_sweep") or {}).get("reason"), d["roofline"].get("merge_kernel_ms"))
PY

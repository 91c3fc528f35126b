#!/bin/bash
# Run ON THE GPU BOX: bin/papr on the 10 GiB bench workload in /dev/shm with the HBM budget capped at 1 GiB (the
# "file larger than HBM" situation of configs[4]): exact sum with the one-sweep ingest (default: one pass over the
# file), exact sum in two passes (PAPR_ONE_SWEEP=0), tree sum with the one-sweep ingest; stdout against the reference's
# recording; then the same file resident (no cap).
#   gpurun -- 'bash tools/cli_stream_probe.sh > gpurun_out/cli_stream.txt'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
F=/dev/shm/papr_stream_$$.cfile
trap 'rm -f "$F"' EXIT
"$R/oracle/mkcfile" "$F" 1342177280 --spike || exit 1
run() { # label, mode flag, env...
  local label=$1 flag=$2; shift 2
  for K in 1 2 3; do
    local t0=$(date +%s%N)
    env PAPR_STATS=1 "$@" "$R/bin/papr" $flag "$F" > /tmp/out_$$.txt 2> /tmp/err_$$.txt
    local t1=$(date +%s%N)
    local golden=$R/tests/golden/big_spike10g.$([ -n "$flag" ] && echo graph || echo default).txt
    local same=$(cmp -s /tmp/out_$$.txt "$golden" && echo identical || echo DIFFERENT)
    echo "$label ${flag:-default} run $K: $(( (t1 - t0) / 1000000 )) ms, stdout $same; $(grep '^{' /tmp/err_$$.txt | python3 -c 'import sys,json; d=json.loads(sys.stdin.readline()); g=d["gpu0_ingest"]; print("ingest+sweep %.3f s (%.1f GB/s), analysis %.4f s, exact_sum %d, swept %d, resolved %d, redo tiles %d, file passes %d, resident %d" % (d["ingest_pass1_s"], d["ingest_GBps"], d["analysis_s"], d["exact_sum"], d["shards_swept"], d["shards_resolved_from_sweep"], d["exact_redo_tiles"], g["file_passes"], g["resident"]))')"
  done
}
for FLAG in "" "-g"; do
  run "1 GiB budget, exact sum, one-sweep ingest " "$FLAG" PAPR_HBM_BUDGET_MB=1024
  run "1 GiB budget, exact sum, two passes       " "$FLAG" PAPR_HBM_BUDGET_MB=1024 PAPR_ONE_SWEEP=0
  run "1 GiB budget, tree sum, one-sweep ingest  " "$FLAG" PAPR_HBM_BUDGET_MB=1024 PAPR_EXACT_SUM=0
  run "resident, exact sum (default)             " "$FLAG"
done
rm -f /tmp/out_$$.txt /tmp/err_$$.txt

#!/bin/bash
# Run ON THE GPU BOX: `bin/papr` on a 10 GiB file that is NOT in the page cache (local disk, drop_caches before every
# run), read with O_DIRECT through one io_uring (default) and through the reader threads (PAPR_IO_URING=0), alternately;
# then the same file warm (page cache) for comparison.  Prints the PAPR_STATS line of every run.
#   gpurun -- 'bash tools/cold_ingest_probe.sh [GiB] > gpurun_out/cold_ingest.txt'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
GIB=${1:-10}
D=${COLD_DIR:-/var/tmp}
F=$D/papr_cold_$$.cfile
N=$((GIB * 134217728))
trap 'rm -f "$F"' EXIT
df -h "$D" | tail -1
"$R/oracle/mkcfile" "$F" $N --spike || exit 1
sync
drop() { sync; echo 3 > /proc/sys/vm/drop_caches 2>/dev/null || echo "(drop_caches not permitted)"; }
for ROUND in 1 2 3; do
  for U in 1 0; do
    drop
    echo "round $ROUND cold PAPR_IO_URING=$U"
    PAPR_STATS=1 PAPR_IO_URING=$U PAPR_EXACT_SUM=0 "$R/bin/papr" "$F" 2>&1 >/dev/null | grep '^{' | head -1
  done
done
echo "warm (page cache)"
cat "$F" > /dev/null
for K in 1 2; do PAPR_STATS=1 PAPR_EXACT_SUM=0 "$R/bin/papr" "$F" 2>&1 >/dev/null | grep '^{' | head -1; done

#!/bin/bash
# Run ON THE GPU BOX (through gpurun): `bin/papr` on a 10 GiB file of the bench workload in /dev/shm, alternating the two
# forms of the ingest's H2D leg (PAPR_H2D=copy: hipMemcpyAsync, =pull: papr_pull_kernel), with the wall-clock marks of
# papr_hip_open (PAPR_OPEN_TRACE) and of the load (PAPR_HOST_TRACE) and the PAPR_STATS line of every run.
#   gpurun -- 'bash tools/ingest_trace.sh [rounds] > gpurun_out/ingest_trace.txt'
set -u
ROUNDS=${1:-4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d)
F=/dev/shm/papr_ingest_trace.cfile
N=$((10*1024*1024*1024/8)); PER=$((N/16))
( for p in $(seq 0 15); do $R/oracle/mkcfile $F $N --spike --part $((p*PER)) $PER & done; wait )
cat $F > /dev/null; cat $F > /dev/null   # (the first two reads of a fresh tmpfs file are LRU-bound)
$R/bin/papr $F > $T/ref.txt 2>/dev/null
for r in $(seq 1 $ROUNDS); do
  for form in copy pull; do
    sleep 0.7; s=$(date +%s.%N)
    PAPR_H2D=$form PAPR_OPEN_TRACE=1 PAPR_HOST_TRACE=1 PAPR_STATS=1 $R/bin/papr $F 2> $T/err.txt > $T/out.txt
    e=$(date +%s.%N)
    echo "== $form wall $(python3 -c "print(round($e-$s,4))") stdout $(cmp -s $T/out.txt $T/ref.txt && echo same || echo DIFFERENT)"
    grep -v '^{' $T/err.txt | cut -c1-400
    grep '^{' $T/err.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('open_s', 'ingest_pass1_s', 'analysis_s', 'total_s', 'ingest_GBps')}, d['gpu0_ingest'])"
  done
done
rm -rf $F $T

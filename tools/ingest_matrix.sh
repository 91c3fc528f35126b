#!/bin/bash
# Ingest tuning matrix: bin/papr on the 10 GiB spike workload in /dev/shm, chunk size x reader threads
# (PAPR_STATS=1 line on stderr: ingest GB/s, seconds)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/ingest; mkdir -p $O
F=/dev/shm/papr_ingest_matrix.cfile
MK=$(ls oracle/_ref/mkcfile oracle/mkcfile 2>/dev/null | head -1)
$MK $F 1342177280 --spike || exit 1
PAPR_STATS=1 bin/papr $F > /dev/null 2> /dev/null < /dev/null   # runtime + page cache warm
for CH in 8 16 32 64; do
  for TH in 8 16 24 32; do
    for REP in 1 2; do
      PAPR_STATS=1 PAPR_CHUNK_MB=$CH PAPR_READ_THREADS=$TH bin/papr $F 2> $O/err.txt > /dev/null < /dev/null
      python - "$CH" "$TH" $O/err.txt <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().splitlines()[-1])
    print("chunk %2s MiB threads %2s: ingest %.2f GB/s  ingest+pass1 %.4f s  open %.3f s  analysis %.4f s" % (sys.argv[1], sys.argv[2], d["ingest_GBps"], d["ingest_pass1_s"], d["open_s"], d["analysis_s"]))
except Exception as e:
    print("failed", e)
PY
    done
  done
done
rm -f $F

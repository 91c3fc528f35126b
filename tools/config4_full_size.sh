#!/bin/bash
# Run ON THE GPU BOX (through gpurun; needs ~260 GiB of /dev/shm and ~25 min): BASELINE.json configs[4] AT ITS OWN SIZE — the 256 GiB
# spike stream (34 359 738 368 samples; eight shards of 2^32 samples each) as ONE file, through `bin/papr [-g]` under a 4 GiB HBM
# budget (every shard streams through the sweep once), as eight shards and as one, against the reference program itself run on
# the same file on the box's host cores: oracle/_ref/papr (plain gcc -O2 build of the reference's papr.c) for the default mode,
# oracle/_ref/papr_o3avx2 (the same source, -O3 -mavx2 -ffp-contract=off: the -O2 build needs ~55 min for -g at this size) for
# both — and the oracle's sequential sum beside them.  Writes gpurun_out/config4_256g/{summary.txt, *.txt}.
#   gpurun --timeout 2100 -- 'bash tools/config4_full_size.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/config4_256g; rm -rf "$O"; mkdir -p "$O"
F=/dev/shm/papr_big_spike256g.cfile
N=34359738368; W=64; PER=$((N / W))
cd "$R"
t0=$(date +%s)
( for p in $(seq 0 $((W - 1))); do oracle/mkcfile $F $N --spike --part $((p * PER)) $PER & done; wait )
echo "file: $(stat -c %s $F) bytes written in $(( $(date +%s) - t0 )) s" > $O/summary.txt
# the references, one host core each, in the background
( s=$(date +%s); oracle/_ref/papr $F > $O/ref_default_O2.txt 2> $O/ref_default_O2.err; echo "reference -O2 default: rc $? in $(( $(date +%s) - s )) s" >> $O/ref_times.txt ) &
( s=$(date +%s); oracle/_ref/papr_o3avx2 $F > $O/ref_default_avx2.txt 2>/dev/null; echo "reference -O3 -mavx2 default: rc $? in $(( $(date +%s) - s )) s" >> $O/ref_times.txt ) &
( s=$(date +%s); oracle/_ref/papr_o3avx2 -g $F > $O/ref_graph_avx2.txt 2>/dev/null; echo "reference -O3 -mavx2 -g: rc $? in $(( $(date +%s) - s )) s" >> $O/ref_times.txt ) &
( s=$(date +%s); python3 -c "
import __graft_entry__ as ge
r = ge.load_oracle().run_file('$F', False)
print(float(r['sum']).hex(), r['n'])" > $O/oracle_sum.txt 2>&1; echo "oracle sequential sum: in $(( $(date +%s) - s )) s" >> $O/ref_times.txt ) &
# the drop-in, while they run
for GPUS in 8 1; do
  for MODE in default graph; do
    FLAG=""; [ $MODE = graph ] && FLAG="-g"
    s=$(date +%s.%N)
    PAPR_STATS=1 PAPR_HBM_BUDGET_MB=4096 PAPR_GPUS=$GPUS PAPR_OVERSUBSCRIBE=1 bin/papr $FLAG $F > $O/papr_${GPUS}shards_$MODE.txt 2> $O/papr_${GPUS}shards_$MODE.err
    echo "bin/papr $FLAG, $GPUS shard(s), 4 GiB budget: rc $? in $(python3 -c "import time; print(round(time.time() - $s, 2))") s; $(tail -1 $O/papr_${GPUS}shards_$MODE.err | cut -c1-700)" >> $O/summary.txt
  done
done
wait
cat $O/ref_times.txt >> $O/summary.txt
echo "oracle: $(cat $O/oracle_sum.txt)" >> $O/summary.txt
for GPUS in 8 1; do
  cmp -s $O/papr_${GPUS}shards_default.txt $O/ref_default_O2.txt && echo "default, $GPUS shard(s): stdout IDENTICAL to the reference (-O2 build)" >> $O/summary.txt || echo "default, $GPUS shard(s): stdout DIFFERS from the reference" >> $O/summary.txt
  cmp -s $O/papr_${GPUS}shards_graph.txt $O/ref_graph_avx2.txt && echo "-g, $GPUS shard(s): stdout IDENTICAL to the reference (-O3 -mavx2 build)" >> $O/summary.txt || echo "-g, $GPUS shard(s): stdout DIFFERS from the reference" >> $O/summary.txt
done
cmp -s $O/ref_default_O2.txt $O/ref_default_avx2.txt && echo "the two builds of the reference agree on the default mode" >> $O/summary.txt || echo "the two builds of the reference DISAGREE on the default mode" >> $O/summary.txt
sha256sum $O/ref_default_O2.txt $O/ref_graph_avx2.txt | sed "s#$O/##" >> $O/summary.txt
rm -f $F
cat $O/summary.txt

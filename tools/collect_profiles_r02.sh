#!/bin/bash
# Run ON THE GPU BOX (through gpurun): round-2 evidence — bench lines, rocprofv3 kernel-trace stats of the same
# commands, and the HBM traffic counters (FETCH_SIZE / WRITE_SIZE, each --pmc set in a pass of its own with
# --kernel-trace only).  Writes gpurun_out/<tag>/; tools/summarize_profiles_r02.py distils profiles/r02_*.
#   gpurun -- 'bash tools/collect_profiles_r02.sh r02'
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
$B                         > $O/bench_default.json 2> $O/bench_default.err
$B --exact --no-cpu-baseline > $O/bench_exact.json 2> $O/bench_exact.err
$B --two-pass --no-cpu-baseline > $O/bench_twopass.json 2> $O/bench_twopass.err
$B --exact --exact-two-pass --no-cpu-baseline > $O/bench_exact_twopass.json 2> $O/bench_exact_twopass.err
$B --workload ts           > $O/bench_ts.json 2> $O/bench_ts.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    $R/bench.py --gpus 1 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
for SIG in bursty constant; do $B --signal $SIG --no-cpu-baseline > $O/bench_$SIG.json 2> $O/bench_$SIG.err; done
$B --force-miss --no-cpu-baseline > $O/bench_miss.json 2> $O/bench_miss.err
for RUN in "default:" "exact:--exact" "ts:--workload ts"; do
  NAME=${RUN%%:*}; FLAGS=${RUN#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$NAME -- \
      $B --steps 20 --warmup 3 --no-cpu-baseline $FLAGS > $O/stats_$NAME.json 2> $O/stats_$NAME.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$NAME -- \
      $B --steps 3 --warmup 1 --no-cpu-baseline $FLAGS > $O/pmc_fetch_$NAME.json 2> $O/pmc_fetch_$NAME.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$NAME -- \
      $B --steps 3 --warmup 1 --no-cpu-baseline $FLAGS > $O/pmc_write_$NAME.json 2> $O/pmc_write_$NAME.err
done
# the TS scan's memory-side requests: one per 128-byte header line, no L2 reuse (what its roofline is priced on)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/pmc_tcc_ts -- \
    $B --workload ts --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_tcc_ts.json 2> $O/pmc_tcc_ts.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_hit_ts -- \
    $B --workload ts --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_hit_ts.json 2> $O/pmc_hit_ts.err
find $O -name "*.csv" -size +8M -delete
ls $O

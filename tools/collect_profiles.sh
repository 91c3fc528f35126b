#!/bin/bash
# Run ON THE GPU BOX (through gpurun): a round's evidence — bench lines, rocprofv3 kernel-trace stats of the same
# commands, and the HBM traffic counters (FETCH_SIZE / WRITE_SIZE, each --pmc set in a pass of its own with
# --kernel-trace only).  Writes gpurun_out/<tag>/; tools/summarize_profiles.py distils profiles/<tag>_*.
#   gpurun -- 'bash tools/collect_profiles.sh r04'
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
H="--headline-only --no-cpu-baseline --no-e2e"
( time $B > $O/bench_full.json 2> $O/bench_full.err ) 2> $O/bench_full.time     # the driver's command: every leg in one line
$B $H                                > $O/bench_default.json 2> $O/bench_default.err
$B $H --exact                        > $O/bench_exact.json 2> $O/bench_exact.err
$B $H --two-pass                     > $O/bench_twopass.json 2> $O/bench_twopass.err
$B --workload ts                     > $O/bench_ts.json 2> $O/bench_ts.err
$B --workload ts --damage 1e-4       > $O/bench_ts_damage.json 2> $O/bench_ts_damage.err
$B --workload ts --damage 1e-3       > $O/bench_ts_damage_1e-3.json 2> $O/bench_ts_damage_1e-3.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    $R/bench.py --gpus 1 $H > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 \
    $R/bench.py --gpus 1 $H --control nccl > $O/bench_torchrun1_nccl_control.json 2> $O/bench_torchrun1_nccl_control.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29536 \
    $R/bench.py --gpus 1 $H --exact > $O/bench_torchrun1_exact.json 2> $O/bench_torchrun1_exact.err
PAPR_XCH_IN_STREAM=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 \
    $R/bench.py --gpus 1 $H > $O/bench_torchrun1_hostpath.json 2> $O/bench_torchrun1_hostpath.err
for SIG in bursty constant; do $B $H --signal $SIG > $O/bench_$SIG.json 2> $O/bench_$SIG.err; done
$B $H --force-miss > $O/bench_miss.json 2> $O/bench_miss.err
for RUN in "default:$H" "exact:$H --exact" "ts:--workload ts --no-cpu-baseline"; do
  NAME=${RUN%%:*}; FLAGS=${RUN#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$NAME -- \
      $B --steps 20 --warmup 3 $FLAGS > $O/stats_$NAME.json 2> $O/stats_$NAME.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$NAME -- \
      $B --steps 3 --warmup 1 $FLAGS > $O/pmc_fetch_$NAME.json 2> $O/pmc_fetch_$NAME.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$NAME -- \
      $B --steps 3 --warmup 1 $FLAGS > $O/pmc_write_$NAME.json 2> $O/pmc_write_$NAME.err
done
# the TS scan's memory-side requests: one per 128-byte header line, no L2 reuse (what its roofline is priced on)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/pmc_tcc_ts -- \
    $B --workload ts --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_tcc_ts.json 2> $O/pmc_tcc_ts.err
find $O -name "*.csv" -size +8M -delete
ls $O

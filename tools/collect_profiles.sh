#!/bin/bash
# Run ON THE GPU BOX (through gpurun): a round's evidence — bench lines, rocprofv3 kernel-trace stats of the same
# commands, and the HBM traffic counters (FETCH_SIZE / WRITE_SIZE, each --pmc set in a pass of its own with
# --kernel-trace only).  Every run keeps its stdout line (<run>.line) and its full record (<run>.json).  Writes gpurun_out/<tag>/; tools/summarize_profiles.py distils profiles/<tag>_*.
#   gpurun -- 'bash tools/collect_profiles.sh r04'
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
H="--headline-only --no-cpu-baseline --no-e2e --no-live-traffic"
( time $B --full-json $O/bench_full.json > $O/bench_full.line 2> $O/bench_full.err ) 2> $O/bench_full.time     # the driver's command: every leg in one line
$B $H                                --full-json $O/bench_default.json > $O/bench_default.line 2> $O/bench_default.err
$B $H --exact                        --full-json $O/bench_exact.json > $O/bench_exact.line 2> $O/bench_exact.err
$B $H --two-pass                     --full-json $O/bench_twopass.json > $O/bench_twopass.line 2> $O/bench_twopass.err
$B --workload ts                     --full-json $O/bench_ts.json > $O/bench_ts.line 2> $O/bench_ts.err
$B --workload ts --damage 1e-4       --full-json $O/bench_ts_damage.json > $O/bench_ts_damage.line 2> $O/bench_ts_damage.err
$B --workload ts --damage 3e-4       --full-json $O/bench_ts_damage_3e-4.json > $O/bench_ts_damage_3e-4.line 2> $O/bench_ts_damage_3e-4.err
$B --workload ts --damage 1e-3       --full-json $O/bench_ts_damage_1e-3.json > $O/bench_ts_damage_1e-3.line 2> $O/bench_ts_damage_1e-3.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    $R/bench.py --gpus 1 $H --full-json $O/bench_torchrun1.json > $O/bench_torchrun1.line 2> $O/bench_torchrun1.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 \
    $R/bench.py --gpus 1 $H --control nccl --full-json $O/bench_torchrun1_nccl_control.json > $O/bench_torchrun1_nccl_control.line 2> $O/bench_torchrun1_nccl_control.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29536 \
    $R/bench.py --gpus 1 $H --exact --full-json $O/bench_torchrun1_exact.json > $O/bench_torchrun1_exact.line 2> $O/bench_torchrun1_exact.err
PAPR_XCH_IN_STREAM=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 \
    $R/bench.py --gpus 1 $H --full-json $O/bench_torchrun1_hostpath.json > $O/bench_torchrun1_hostpath.line 2> $O/bench_torchrun1_hostpath.err
# `python bench.py --gpus 2` with no launcher around it: bench.py starts its own ranks (here two ranks share the one GPU over gloo: the code path
# and the line's shape — weak + strong members — not a measurement)
$B --gpus 2 --backend gloo --steps 10 --warmup 2 --no-cpu-baseline --full-json $O/bench_selflaunch_n2_gloo.json > $O/bench_selflaunch_n2_gloo.line 2> $O/bench_selflaunch_n2_gloo.err
for SIG in bursty constant; do $B $H --signal $SIG --full-json $O/bench_$SIG.json > $O/bench_$SIG.line 2> $O/bench_$SIG.err; done
$B $H --force-miss --full-json $O/bench_miss.json > $O/bench_miss.line 2> $O/bench_miss.err
for RUN in "default:$H" "exact:$H --exact" "ts:--workload ts --no-cpu-baseline"; do
  NAME=${RUN%%:*}; FLAGS=${RUN#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$NAME -- \
      $B --steps 20 --warmup 3 $FLAGS --full-json $O/stats_$NAME.json > $O/stats_$NAME.line 2> $O/stats_$NAME.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$NAME -- \
      $B --steps 3 --warmup 1 $FLAGS --full-json $O/pmc_fetch_$NAME.json > $O/pmc_fetch_$NAME.line 2> $O/pmc_fetch_$NAME.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$NAME -- \
      $B --steps 3 --warmup 1 $FLAGS --full-json $O/pmc_write_$NAME.json > $O/pmc_write_$NAME.line 2> $O/pmc_write_$NAME.err
done
# the TS scan's memory-side requests: one per 128-byte header line, no L2 reuse (what its roofline is priced on)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/pmc_tcc_ts -- \
    $B --workload ts --steps 2 --warmup 1 --no-cpu-baseline --full-json $O/pmc_tcc_ts.json > $O/pmc_tcc_ts.line 2> $O/pmc_tcc_ts.err
find $O -name "*.csv" -size +8M -delete
ls $O

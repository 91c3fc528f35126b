#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats of the bench
# command, the PMC passes (each on its own, with --kernel-trace only), the HBM read
# probe and plain bench lines.  Writes gpurun_out/<tag>/; tools/summarize_profiles.py
# then distils what is committed under profiles/.
#   gpurun -- 'bash tools/collect_profiles.sh r01'
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
$R/bin/hbm_read_probe 10 10 > $O/hbm_read_probe.txt 2>&1
$B                > $O/bench_default.json 2> $O/bench_default.err
$B --mode graph   > $O/bench_graph.json   2> $O/bench_graph.err
$B --two-pass              --no-cpu-baseline > $O/bench_twopass_default.json 2> $O/bench_twopass_default.err
$B --two-pass --mode graph --no-cpu-baseline > $O/bench_twopass_graph.json   2> $O/bench_twopass_graph.err
$B --exact --no-cpu-baseline > $O/bench_exact.json 2> $O/bench_exact.err
$B --exact --mode graph --no-cpu-baseline > $O/bench_exact_graph.json 2> $O/bench_exact_graph.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_exact -- \
    $B --steps 20 --warmup 3 --exact --no-cpu-baseline > $O/stats_exact.json 2> $O/stats_exact.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    $R/bench.py --gpus 1 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
# end-to-end CLI (file -> pinned host -> HBM), PCIe-inclusive
$R/oracle/mkcfile /dev/shm/papr_prof_10g.cfile 1342177280 --spike
for i in 1 2 3; do
  PAPR_STATS=1 $R/bin/papr /dev/shm/papr_prof_10g.cfile 2>> $O/cli_e2e.txt > $O/cli_default.txt
  PAPR_STATS=1 $R/bin/papr -g /dev/shm/papr_prof_10g.cfile 2>> $O/cli_e2e.txt > $O/cli_graph.txt
done
PAPR_EXACT_SUM=0 PAPR_STATS=1 $R/bin/papr /dev/shm/papr_prof_10g.cfile 2>> $O/cli_e2e.txt > /dev/null
cmp $O/cli_default.txt $R/tests/golden/big_spike10g.default.txt && cmp $O/cli_graph.txt $R/tests/golden/big_spike10g.graph.txt \
  && echo "CLI stdout identical to the reference on the 10 GiB workload (both modes)" >> $O/cli_e2e.txt
# the file does not fit the HBM budget (configs[4] situation): streamed through device staging
echo "--- PAPR_HBM_BUDGET_MB=1024: exact sum (two reads of the file) / tree sum, one-sweep ingest (one read) / tree sum, two passes" >> $O/cli_e2e.txt
for i in 1 2 3; do
  for M in "" "-g"; do
    PAPR_HBM_BUDGET_MB=1024 PAPR_STATS=1 $R/bin/papr $M /dev/shm/papr_prof_10g.cfile 2>> $O/cli_e2e.txt > $O/cli_streamed_exact$M.txt
    PAPR_HBM_BUDGET_MB=1024 PAPR_EXACT_SUM=0 PAPR_STATS=1 $R/bin/papr $M /dev/shm/papr_prof_10g.cfile 2>> $O/cli_e2e.txt > $O/cli_streamed_sweep$M.txt
    PAPR_HBM_BUDGET_MB=1024 PAPR_EXACT_SUM=0 PAPR_ONE_SWEEP=0 PAPR_STATS=1 $R/bin/papr $M /dev/shm/papr_prof_10g.cfile 2>> $O/cli_e2e.txt > $O/cli_streamed_twopass$M.txt
  done
done
cmp $O/cli_streamed_exact.txt $R/tests/golden/big_spike10g.default.txt && cmp $O/cli_streamed_exact-g.txt $R/tests/golden/big_spike10g.graph.txt \
  && cmp $O/cli_streamed_sweep.txt $R/tests/golden/big_spike10g.default.txt && cmp $O/cli_streamed_sweep-g.txt $R/tests/golden/big_spike10g.graph.txt \
  && cmp $O/cli_streamed_twopass.txt $R/tests/golden/big_spike10g.default.txt \
  && echo "streamed: stdout identical to the reference in all three forms (both modes)" >> $O/cli_e2e.txt
rm -f /dev/shm/papr_prof_10g.cfile
# one-sweep (the default) and two-pass, both level tables: kernel-trace stats, then each PMC set in a pass of its own
for MODE in default graph; do
  for WAY in "" twopass_; do
    FLAG=""; [ -n "$WAY" ] && FLAG="--two-pass"
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$WAY$MODE -- \
        $B --steps 20 --warmup 3 --mode $MODE $FLAG --no-cpu-baseline > $O/stats_$WAY$MODE.json 2> $O/stats_$WAY$MODE.err
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$WAY$MODE -- \
        $B --steps 3 --warmup 1 --mode $MODE $FLAG --no-cpu-baseline > $O/pmc_fetch_$WAY$MODE.json 2> $O/pmc_fetch_$WAY$MODE.err
  done
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$MODE -- \
      $B --steps 3 --warmup 1 --mode $MODE --no-cpu-baseline > $O/pmc_write_$MODE.json 2> $O/pmc_write_$MODE.err
done
for WAY in "" twopass_; do
  FLAG=""; [ -n "$WAY" ] && FLAG="--two-pass"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv \
      -d $O/pmc_tcc_$WAY -- $B --steps 3 --warmup 1 --mode graph $FLAG --no-cpu-baseline > $O/pmc_tcc_$WAY.json 2> $O/pmc_tcc_$WAY.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS \
      --kernel-trace --output-format csv -d $O/pmc_sq_$WAY -- $B --steps 3 --warmup 1 --mode graph $FLAG --no-cpu-baseline > $O/pmc_sq_$WAY.json 2> $O/pmc_sq_$WAY.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
      --kernel-trace --output-format csv -d $O/pmc_lds_$WAY -- $B --steps 3 --warmup 1 --mode graph $FLAG --no-cpu-baseline > $O/pmc_lds_$WAY.json 2> $O/pmc_lds_$WAY.err
done
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv \
    -d $O/pmc_grbm -- $B --steps 3 --warmup 1 --mode graph --no-cpu-baseline > $O/pmc_grbm.json 2> $O/pmc_grbm.err
ls $O

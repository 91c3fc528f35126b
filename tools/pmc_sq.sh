#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ counters of the two product sweep kernels, both tables, and of the packet scan, one --pmc set per
# rocprofv3 pass (with --kernel-trace only).  tools/pmc_summarize.py prints the per-kernel averages into
# gpurun_out/<tag>/summary.txt; copy that to profiles/<tag>_pmc_sq_product_kernels.txt.
#   gpurun -- 'bash tools/pmc_sq.sh pmc_r03'
set -u
TAG=${1:-pmc_r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
declare -A SETS
SETS[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
SETS[b]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
SETS[c]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC"
for RUN in ${PMC_RUNS:-"sweep_default:default:" "sweep_graph:graph:" "sweep3_default:default:--exact" "sweep3_graph:graph:--exact"}; do
  IFS=: read NAME MODE EXTRA <<< "$RUN"
  mkdir -p $O/$NAME
  for s in a b c; do
    rocprofv3 --pmc ${SETS[$s]} --kernel-trace --output-format csv -d $O/$NAME/pmc_$s -- \
        python $R/bench.py --headline-only --mode $MODE --steps 3 --warmup 1 --no-cpu-baseline --no-e2e $EXTRA > $O/$NAME/pmc_$s.json 2> $O/$NAME/pmc_$s.err
  done
  echo "==== $NAME (python bench.py --headline-only --mode $MODE $EXTRA; per 10 GiB launch)" >> $O/summary.txt
  python $R/tools/pmc_summarize.py $O/$NAME 2>&1 | awk '/^papr_sweep/{p=1} /^papr_(stats|ccdf|exact|est|guess|true)/{p=0} p' >> $O/summary.txt
done
# the transport-stream scan (clean stream, and damaged at 1e-3)
for RUN in ${PMC_TS_RUNS-"ts_clean:" "ts_damage_1e-3:--damage 1e-3"}; do
  IFS=: read NAME EXTRA <<< "$RUN"
  mkdir -p $O/$NAME
  for s in a b c; do
    rocprofv3 --pmc ${SETS[$s]} --kernel-trace --output-format csv -d $O/$NAME/pmc_$s -- \
        python $R/bench.py --workload ts --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $O/$NAME/pmc_$s.json 2> $O/$NAME/pmc_$s.err
  done
  echo "==== $NAME (python bench.py --workload ts $EXTRA; per 10 GiB scan)" >> $O/summary.txt
  python $R/tools/pmc_summarize.py $O/$NAME 2>&1 | awk '/^ts_scan/{p=1} /^ts_(merge|reset|gen)/{p=0} p' >> $O/summary.txt
done
find $O -name "*.csv" -delete
cat $O/summary.txt | head -200

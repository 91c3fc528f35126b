#!/bin/bash
# Run ON THE GPU BOX: hardware counters of the one-sweep kernels next to the two-pass kernels, one --pmc set per
# rocprofv3 pass (with --kernel-trace only), all on tools/sweep_probe.py.  Output: gpurun_out/<tag>/pmc_<set>/ ;
# tools/pmc_summarize.py prints the per-kernel averages.
#   gpurun -- 'bash tools/pmc_probe.sh TAG "4:1024,32:256" graph'
set -u
TAG=${1:-pmc}; CONFIGS=${2:-4:1024,32:256}; MODES=${3:-graph}; EXTRA=${4:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
declare -A SETS
SETS[a]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
SETS[b]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
SETS[c]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC"
SETS[d]="TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
SETS[e]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
SETS[f]="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum"
SETS[g]="GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES SQ_LEVEL_WAVES"
for s in ${PMC_SETS:-a b c d e f g}; do
  rocprofv3 --pmc ${SETS[$s]} --kernel-trace --output-format csv -d $O/pmc_$s -- \
      python $R/tools/sweep_probe.py --reps 3 --bands 14 --configs $CONFIGS --modes $MODES $EXTRA > $O/pmc_$s.txt 2> $O/pmc_$s.err
  tail -2 $O/pmc_$s.err | cut -c1-200
done
python $R/tools/pmc_summarize.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
# keep only the summaries (the raw CSVs are large)
find $O -name "*.csv" -size +2M -delete

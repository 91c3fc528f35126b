#!/bin/bash
# Re-run just the plain bench lines (after tools/summarize_profiles.py has refreshed profiles/pmc_traffic.json, so
# that roofline.traffic in the committed lines comes from the same round's PMC pass).
#   gpurun -- 'bash tools/collect_bench_lines.sh r01'
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
$B                > $O/bench_default.json 2> $O/bench_default.err
$B --mode graph   > $O/bench_graph.json   2> $O/bench_graph.err
$B --two-pass              --no-cpu-baseline > $O/bench_twopass_default.json 2> $O/bench_twopass_default.err
$B --two-pass --mode graph --no-cpu-baseline > $O/bench_twopass_graph.json   2> $O/bench_twopass_graph.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 \
    $R/bench.py --gpus 1 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err

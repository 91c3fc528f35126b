#!/bin/bash
# Run ON THE GPU BOX (through gpurun): ONE library under several settings of the environment, alternating fresh processes,
# `bench.py --headline-only` with the given flags; prints kernel / step time of both tables per setting and round.
#   gpurun -- 'bash tools/env_ab.sh 4 PAPR_XCD_SKEW=0 PAPR_XCD_SKEW=48 PAPR_XCD_SKEW=24 -- --exact'
# AB_TORCHRUN=1: every process as the driver launches ranks (torch.distributed.run, one rank: RCCL resident beside the step).
set -u
ROUNDS=$1; shift
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ $# -gt 0 ] && shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for i in $(seq 1 "$ROUNDS"); do
  for S in "${SETS[@]}"; do
    PY="python3"; [ "${AB_TORCHRUN:-0}" = 1 ] && PY="python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571"
    env $S $PY bench.py --headline-only --no-cpu-baseline --no-e2e --steps 20 --warmup 5 "$@" 2>/dev/null |
      python3 -c "
import json, sys
b = json.loads(sys.stdin.readline()); r = b['roofline']; g = (r.get('legs') or {}).get('graph') or {}
print('%-22s' % '$S', 'round $i', 'default: kernel %.4f step %.4f' % (r['kernel_ms'], b['ms_per_step']),
      '| -g: kernel %.4f step %.4f' % (g.get('kernel_ms', -1), g.get('ms_per_step', -1)), '| parity', b.get('parity_in_run'), g.get('parity_in_run'))
"
  done
done

#!/bin/bash
# Run ON THE GPU BOX (through gpurun): N builds of libpaprhip.so against each other on one box, alternating fresh
# processes (PAPR_LIB_PATH), `bench.py --headline-only` with the given flags; prints kernel / step time of both tables.
#   gpurun -- 'bash tools/lib_abn.sh 3 scratch/ab/a.so scratch/ab/b.so scratch/ab/c.so -- --exact'
set -u
ROUNDS=$1; shift
LIBS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ $# -gt 0 ] && shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for i in $(seq 1 "$ROUNDS"); do
  for L in "${LIBS[@]}"; do
    PAPR_LIB_PATH=$(realpath "$L") python3 bench.py --headline-only --no-cpu-baseline --no-e2e --steps 20 --warmup 5 "$@" 2>/dev/null |
      python3 -c "
import json, sys
b = json.loads(sys.stdin.readline()); r = b['roofline']; g = (r.get('legs') or {}).get('graph') or {}
print('%-28s' % '$L', 'round $i', 'default: kernel %.4f step %.4f' % (r['kernel_ms'], b['ms_per_step']),
      '| -g: kernel %.4f step %.4f' % (g.get('kernel_ms', -1), g.get('ms_per_step', -1)), '| parity', b.get('parity_in_run'), g.get('parity_in_run'))
"
  done
done

#!/bin/bash
# Run ON THE GPU BOX (through gpurun): two builds of libpaprhip.so against each other on one box, alternating fresh
# processes (PAPR_LIB_PATH), `bench.py --headline-only` with the given flags; prints kernel / step time of both tables.
#   gpurun -- 'bash tools/lib_ab.sh scratch/ab/lib_base.so dtv-utils_amd/libpaprhip.so 3 --exact'
set -u
A=$1; B=$2; ROUNDS=${3:-3}; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for i in $(seq 1 "$ROUNDS"); do
  for L in "$A" "$B"; do
    PAPR_LIB_PATH=$(realpath "$L") python3 bench.py --headline-only --no-cpu-baseline --no-e2e --steps 20 --warmup 5 "$@" 2>/dev/null |
      python3 -c "
import json, sys
b = json.loads(sys.stdin.readline()); r = b['roofline']; g = (r.get('legs') or {}).get('graph') or {}
print('$L', 'round $i', 'default: kernel %.4f step %.4f' % (r['kernel_ms'], b['ms_per_step']),
      '| -g: kernel %.4f step %.4f' % (g.get('kernel_ms', -1), g.get('ms_per_step', -1)), '| parity', b.get('parity_in_run'))
"
  done
done

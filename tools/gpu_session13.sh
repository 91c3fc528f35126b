#!/bin/bash
# GPU session 13: the N-rank bench lines on ONE GPU (gloo transport, ranks share the device): parity_in_run against the
# reference's recorded stdout for the 20 / 40 / 80 GiB streams, plain and --exact
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s13; mkdir -p $O
for N in 2 4 8; do
  for FLAGS in "" "--exact"; do
    T=$(echo "n${N}${FLAGS}" | tr -d ' -')
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
        bench.py --gpus $N --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-e2e $FLAGS > $O/$T.json 2> $O/$T.err
    python - <<PY
import json
try:
    d=json.load(open("$O/$T.json"))
    print("$T", "n_gpus", d["n_gpus"], "parity", d["parity_in_run"], d["parity_golden"], "| -g", d["graph"]["parity_in_run"], d["graph"]["parity_golden"], "| sum", d["config"]["sum_hex"], "ms", round(d["ms_per_step"],3), d["exchange"])
except Exception as e:
    print("$T failed", e); print(open("$O/$T.err").read()[-1500:])
PY
  done
done

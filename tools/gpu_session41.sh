#!/bin/bash
# GPU session 41: full suite + the host (with-peers) path's step under torchrun N=1 after the finalize-copy change
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s41; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 < /dev/null; tail -3 $O/pytest.log | cut -c1-300
line() {
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    g = d["graph"]
    print("%-24s step %.4f / %.4f  kernel %.4f / %.4f  aux %.4f / %.4f host %.4f / %.4f parity %s %s  exch %s" % (
        sys.argv[1], d["ms_per_step"], g["ms_per_step"], d["roofline"]["kernel_ms"], g["roofline"]["kernel_ms"],
        d["kernels"]["estimate_and_recount_kernels"]["ms_per_step"], g["kernels"]["estimate_and_recount_kernels"]["ms_per_step"],
        d["kernels"]["host_and_exchange_ms_per_step"], g["kernels"]["host_and_exchange_ms_per_step"],
        d["parity_in_run"], g["parity_in_run"], json.dumps(d["exchange"])[:160]))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for ROUND in 1 2 3; do
  for FC in 0 1; do
  PAPR_FUSED_COPIES=$FC timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$ROUND \
      bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/t$ROUND.json 2> $O/t.err < /dev/null
  line "r$ROUND torchrun1 copies=$FC" $O/t$ROUND.json
  done
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b$ROUND.json 2> $O/b.err < /dev/null
  line "r$ROUND alone" $O/b$ROUND.json
done

#!/bin/bash
# GPU session 20: exact default = wide-batch ballot kernel, program replay overlapped with the recount
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s20; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for ROUND in 1 2 3; do
  timeout 300 python bench.py --exact --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/b.json 2> $O/b.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b.json"))
    print("round $ROUND --exact: kernel %.4f / -g %.4f  step %.4f / %.4f parity %s %s host %.3f / %.3f" % (d["roofline"]["kernel_ms"], d["graph"]["roofline"]["kernel_ms"], d["ms_per_step"], d["graph"]["ms_per_step"], d["parity_in_run"], d["graph"]["parity_in_run"], d["kernels"]["host_and_exchange_ms_per_step"], d["graph"]["kernels"]["host_and_exchange_ms_per_step"]))
except Exception as e:
    print("failed", e, open("$O/b.err").read()[-600:])
PY
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --exact --steps 50 --warmup 5 --no-cpu-baseline > $O/t.json 2> $O/t.err
python -c "
import json; d=json.load(open('$O/t.json')); print('torchrun --exact: step %.4f / %.4f parity %s %s' % (d['ms_per_step'], d['graph']['ms_per_step'], d['parity_in_run'], d['graph']['parity_in_run']), d['exchange'])"

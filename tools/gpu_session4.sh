#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/s4; mkdir -p $O
timeout 300 python -m pytest tests/test_ts_scan.py tests/test_gpu_parity.py -m gpu -q -x -k "ts_ or scan or arbitrary_level or abi_reproduces" > $O/pytest_ts.log 2>&1; tail -4 $O/pytest_ts.log
timeout 300 python bench.py --workload ts --steps 10 --warmup 2 > $O/bench_ts.json 2> $O/bench_ts.err; cut -c1-1500 $O/bench_ts.json; tail -3 $O/bench_ts.err
timeout 600 python tools/exact_sweep_probe.py --reps 10 --variants 48,49,50,51,52,54 --skip-small > $O/exact_probe.txt 2>&1; grep -v "^n=.*counts_ok=True" $O/exact_probe.txt | cut -c1-250
timeout 900 python tools/sweep_probe.py --reps 10 --bands 14 \
  --configs 4:1024,1:512,1:1024,1:2048,2:512,6:512,6:1024,8:512,8:1024,9:512,9:1024,10:1024,10:2048,11:2048,13:512,13:1024,14:512,15:512,17:256,17:512,0:512,0:1024,3:512,7:1024,7:2048,12:512,12:1024,44:256,45:256,4:1024 \
  > $O/sweep_probe.txt 2>&1; grep "v=\|two-pass" $O/sweep_probe.txt | cut -c1-120
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/s4/bench_torchrun1.json"))
print("torchrun1:", d["ms_per_step"], d["exchange"], d["parity_in_run"], d["graph"]["ms_per_step"], d["graph"]["parity_in_run"])
PY
for SIG in bursty constant; do timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --signal $SIG > $O/bench_$SIG.json 2> $O/bench_$SIG.err; python - <<PY
import json
d=json.load(open("gpurun_out/s4/bench_$SIG.json"))
print("$SIG:", round(d["ms_per_step"],3), d["config"]["one_sweep"], "graph:", round(d["graph"]["ms_per_step"],3), d["graph"]["config"]["one_sweep"])
PY
done
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --force-miss > $O/bench_miss.json 2> $O/bench_miss.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/s4/bench_miss.json"))
print("miss:", round(d["ms_per_step"],3), d["config"]["one_sweep"], "graph:", round(d["graph"]["ms_per_step"],3))
PY
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --exact > $O/bench_exact.json 2> $O/bench_exact.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/s4/bench_exact.json"))
print("exact:", round(d["ms_per_step"],3), d["config"]["sum_hex"], d["parity_in_run"], d["config"]["one_sweep"], "graph:", round(d["graph"]["ms_per_step"],3), d["graph"]["parity_in_run"])
PY

#!/bin/bash
# A/B of library builds (PAPR_LIB_PATH): exact one-read sweep parity at 10 GiB and 0.5 GiB, constant envelope
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=gpurun_out/bisect; mkdir -p $O
for L in ""; do
  N=$(basename "${L:-default}")
  for G in 10 0.5; do
    for SIG in gauss constant; do
    PAPR_LIB_PATH=${L:+$R/$L} timeout 300 python bench.py --exact --gib $G --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --signal $SIG > $O/b_${N}_$G.json 2> $O/b_${N}_$G.err
    python - <<PY
import json
try:
    d=json.load(open("$O/b_${N}_$G.json"))
    print("$N", "$G", "$SIG", d["parity_in_run"], d["graph"]["parity_in_run"], d["config"]["counts_crc32"], d["graph"]["config"]["counts_crc32"], d["config"]["one_sweep"]["reason"], d["config"]["one_sweep"]["gave_up"], d["graph"]["config"]["one_sweep"]["gave_up"], round(d["roofline"]["kernel_ms"],4), round(d["ms_per_step"],4))
except Exception as e:
    print("$N", "$G", "failed", e)
PY
    done
  done
done
for SIG in gauss constant; do
timeout 300 python bench.py --exact --exact-two-pass --gib 0.5 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --signal $SIG > $O/b_two.json 2>$O/b_two.err
python -c "
import json; d=json.load(open('$O/b_two.json')); print('two-pass 0.5 $SIG', d['config']['counts_crc32'], d['graph']['config']['counts_crc32'])"
done

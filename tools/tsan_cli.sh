#!/bin/bash
# bin/papr with its host code under ThreadSanitizer (the reader threads of the ingest, the per-shard threads and their hub, the
# packet scan's line pool): build here (`tools/tsan_cli.sh build`: cross-compiles, no GPU), run ON THE GPU BOX
# (`gpurun -- 'bash tools/tsan_cli.sh run'`).  The HIP / HSA runtimes are not instrumented and report races of their own (their
# internal threads against API calls): tools/tsan_filter.py counts the reports in which one of the two accesses is in OUR code.
#   resident exact | streamed -g | three shards on one GPU over the in-process hub | three shards streamed | the packet scan
#   (tests/c/ts_scan_harness.cpp through the C ABI: a damaged stream scanned again and again, every report the first one's)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
C=dtv-utils_amd/csrc
if [ "${1:-run}" = build ]; then
  make lib >/dev/null || exit 1
  mkdir -p scratch/tsan bin
  FL="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$C -I/opt/rocm/include -Xarch_host -fsanitize=thread"
  for f in papr_runtime papr_ingest papr_sweep_rt papr_exact_rt ts_runtime papr_exchange papr_analyze; do
    /opt/rocm/bin/hipcc $FL -c $C/$f.cpp -o scratch/tsan/$f.o || exit 1
  done
  CL=/opt/rocm/lib/llvm/bin/clang
  $CL -O1 -g -fPIC -ffp-contract=off -fsanitize=thread -Iinclude -I$C -c $C/papr_host.c -o scratch/tsan/papr_host.o || exit 1
  $CL -O1 -g -fPIC -ffp-contract=off -fsanitize=thread -Iinclude -I$C -c $C/ts_host.c -o scratch/tsan/ts_host.o || exit 1
  $CL -O1 -g -ffp-contract=off -fsanitize=thread -Iinclude -c dtv-utils_amd/host/papr_main.c -o scratch/tsan/papr_main.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=thread scratch/tsan/*.o $C/papr_kernels.o $C/papr_sweep.o $C/papr_exact.o $C/ts_kernels.o \
      -o bin/papr_tsan -lm -lpthread -ldl || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -Iinclude -Xarch_host -fsanitize=thread -c tests/c/ts_scan_harness.cpp -o scratch/tsan/ts_scan_harness.o2 || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=thread scratch/tsan/ts_scan_harness.o2 $(ls scratch/tsan/*.o | grep -v papr_main.o) \
      $C/papr_kernels.o $C/papr_sweep.o $C/papr_exact.o $C/ts_kernels.o -o bin/ts_tsan -lm -lpthread -ldl || exit 1
  echo "built bin/papr_tsan bin/ts_tsan"
  exit 0
fi
F=/dev/shm/tsan.cfile
oracle/mkcfile $F 300000007 --spike --extra-floats 1 >/dev/null
ref=$(bin/papr $F | md5sum); refg=$(bin/papr -g $F | md5sum)
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4"
run() {
  name=$1; want=$2; shift 2
  timeout 300 env PAPR_TEARDOWN=1 "$@" > /tmp/tsan_out.txt 2> /tmp/tsan_err.txt
  got=$(md5sum < /tmp/tsan_out.txt)
  echo "=== $name: stdout $([ "$got" = "$want" ] && echo identical to bin/papr\'s || echo DIFFERS)"
  python3 tools/tsan_filter.py /tmp/tsan_err.txt
}
run "resident, exact sum" "$ref" bin/papr_tsan $F
run "streamed (512 MiB of HBM), -g" "$refg" PAPR_HBM_BUDGET_MB=512 bin/papr_tsan -g $F
run "three shards on one GPU, in-process hub" "$ref" PAPR_GPUS=3 PAPR_OVERSUBSCRIBE=1 PAPR_XCH=threads bin/papr_tsan $F
run "three shards on one GPU, RCCL refused (one device): hub" "$ref" PAPR_GPUS=3 PAPR_OVERSUBSCRIBE=1 bin/papr_tsan $F
run "three shards, streamed, -g" "$refg" PAPR_GPUS=3 PAPR_OVERSUBSCRIBE=1 PAPR_HBM_BUDGET_MB=256 bin/papr_tsan -g $F
# round 6: the communicators coming up in threads of their own beside the ingest (papr_exchange_open_rccl_local_async / _adopt_rccl) —
# taken, failed, late, and three shards of which one fails while the others sit in ncclCommInitRank — and a stream through a FIFO
run "one shard, RCCL taken when the shard is loaded" "$ref" PAPR_GPUS=1 PAPR_XCH=rccl bin/papr_tsan $F
run "one shard, RCCL set-up fails: hub" "$ref" PAPR_GPUS=1 PAPR_XCH=rccl PAPR_XCH_BIND_FAIL=all bin/papr_tsan $F
run "one shard, RCCL late (auto): hub" "$ref" PAPR_GPUS=1 PAPR_XCH=auto PAPR_XCH_BIND_DELAY_MS=20000 bin/papr_tsan $F
run "three shards, one set-up fails, the others abandoned" "$ref" PAPR_GPUS=3 PAPR_OVERSUBSCRIBE=1 PAPR_XCH_BIND_SHARED_OK=1 PAPR_XCH_BIND_FAIL=1 PAPR_XCH_BIND_TIMEOUT_S=5 bin/papr_tsan $F
rm -f /tmp/tsan.fifo; mkfifo /tmp/tsan.fifo
( cat $F > /tmp/tsan.fifo & ) ; reff=$(oracle/_ref/papr /tmp/tsan.fifo 2>/dev/null | md5sum)
( cat $F > /tmp/tsan.fifo & ) ; run "a stream through a FIFO, windows of 64 MiB" "$reff" PAPR_STREAM_WINDOW_MB=64 bin/papr_tsan /tmp/tsan.fifo
rm -f /tmp/tsan.fifo
rm -f $F
# the packet scan: a damaged stream's report lines are laid out by the host's line pool (eight spinning threads), scan after scan
for SPEC in "8000000 500 12" "57000000 1000 6"; do
  timeout 300 env TS_SCAN_FORM=auto bin/ts_tsan $SPEC > /tmp/tsan_out.txt 2> /tmp/tsan_err.txt
  echo "=== packet scan, $(cat /tmp/tsan_out.txt)"
  python3 tools/tsan_filter.py /tmp/tsan_err.txt
done

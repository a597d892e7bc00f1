#!/bin/bash
# Run the CPU tests of the native host code under a sanitizer build of the
# library (CPU only: the GPU pool has no sanitizer support).
#   tools/run_sanitized.sh thread               # ThreadSanitizer
#   tools/run_sanitized.sh address,undefined    # ASan + UBSan
# Extra arguments go to pytest.  (The gloo rank harness and the two tests that
# run `bench.py --gpus 2` over gloo are left out: PyTorch's own ProcessGroupGloo
# reports races under TSan.)  The full 10^5-job team stress:
#   BNPC_STRESS_JOBS=100000 tools/run_sanitized.sh thread -k team
set -e
SAN=${1:-thread}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
case "$SAN" in
  thread) PRE=$RT/libclang_rt.tsan-x86_64.so ;;
  *) PRE=$RT/libclang_rt.asan-x86_64.so ;;
esac
BNPC_SANITIZE=$SAN python3 -m bnpc_amd.build
export BNPC_LIB=$ROOT/build/libbnpc_hip.${SAN//,/_}.so
# Python itself is not instrumented: report races in the library only
LOGDIR=${SAN_LOG_DIR:-/tmp/bnpc_sanitizer}
rm -rf "$LOGDIR"; mkdir -p "$LOGDIR"
export TSAN_OPTIONS="log_path=$LOGDIR/tsan halt_on_error=0 report_signal_unsafe=0 die_after_fork=0 ${TSAN_OPTIONS}"
export ASAN_OPTIONS="detect_leaks=0 ${ASAN_OPTIONS}"
cd "$ROOT"
LD_PRELOAD=$PRE python3 -m pytest tests/test_native_sweeps.py \
    tests/test_host_logic.py tests/test_multichain.py tests/test_fastdist.py \
    --deselect tests/test_multichain.py::test_bench_rank_harness_gloo_world2 \
    --deselect tests/test_multichain.py::test_bench_command_with_gpus_2 \
    --deselect tests/test_multichain.py::test_bench_command_reports_a_rank_that_fails \
    -q -p no:cacheprovider "$@" || RC=$?
echo "sanitizer reports in $LOGDIR: $(ls "$LOGDIR" | wc -l) file(s)"
cat "$LOGDIR"/* 2>/dev/null | grep -E "^(WARNING|SUMMARY)" | sort | uniq -c
exit ${RC:-0}

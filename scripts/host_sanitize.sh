#!/bin/bash
# Sanitizer gate of the native host stage (csrc/nrv_host.c + csrc/nrv_host_fast5.c): builds
# tools/hostfuzz/host_fuzz.c with AddressSanitizer + UndefinedBehaviorSanitizer (no recovery) and with ThreadSanitizer and runs
#   api      argument edge cases of every exported entry point
#   fuzz     N mutations per fixture file (metadata flips, extreme 2/4/8-byte values in header messages, B-tree / heap nodes
#            with cycles, the Events compound type, ROWS of the Events table, truncations, chunk keys) through the image
#            parser, the file + bundle entry points and the finishers
#   threads  8 threads loading bundles and finishing reads at once (ThreadSanitizer)
# Any sanitizer report aborts the run: exit code != 0.   usage: host_sanitize.sh [N_MUTATIONS_PER_FILE=1000] [SEED=1]
set -u
N=${1:-1000}; SEED=${2:-1}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d "${TMPDIR:-/tmp}/nrv_sanitize.XXXXXX")
trap 'rm -rf "$W"' EXIT
FILES=$(ls "$ROOT"/tests/golden/fast5/*.fast5 "$ROOT"/tests/golden/fast5_more/*.fast5)
CF="-O1 -g -fno-omit-frame-pointer -fno-fast-math -ffp-contract=off -pthread"
gcc $CF -fsanitize=address,undefined -fno-sanitize-recover=all -o "$W/host_fuzz_asan" "$ROOT/tools/hostfuzz/host_fuzz.c" -lm -lz -ldl || exit 2
export ASAN_OPTIONS=allocator_may_return_null=1:detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
"$W/host_fuzz_asan" api "$W" $FILES || { echo "SANITIZE FAIL: api"; exit 1; }
"$W/host_fuzz_asan" fuzz "$N" "$SEED" "$W" $FILES || { echo "SANITIZE FAIL: fuzz"; exit 1; }
if gcc $CF -fsanitize=thread -o "$W/host_fuzz_tsan" "$ROOT/tools/hostfuzz/host_fuzz.c" -lm -lz -ldl 2>"$W/tsan_build.log"; then
  # (TSan needs a fixed address-space layout on some kernels: setarch -R when the first try cannot map its shadow)
  TSAN_OPTIONS=halt_on_error=1:second_deadlock_stack=1 "$W/host_fuzz_tsan" threads 8 6 "$W" $FILES > "$W/tsan.log" 2>&1; rc=$?
  if [ $rc -ne 0 ] && grep -q "unexpected memory mapping\|FATAL: ThreadSanitizer" "$W/tsan.log"; then
    TSAN_OPTIONS=halt_on_error=1 setarch "$(uname -m)" -R "$W/host_fuzz_tsan" threads 8 6 "$W" $FILES > "$W/tsan.log" 2>&1; rc=$?
  fi
  cat "$W/tsan.log"
  [ $rc -eq 0 ] || { echo "SANITIZE FAIL: threads"; exit 1; }
else
  cat "$W/tsan_build.log"; echo "SANITIZE FAIL: the ThreadSanitizer build"; exit 2
fi
echo "SANITIZE OK"

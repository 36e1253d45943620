#!/bin/bash
# Sanitizer build of the C ABI's HOST side (sanitizer run for the native layer) and a GPU test pass
# with it.
#   build only (no GPU needed):  bash tools/sanitize.sh build
#   on the GPU box:              /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/sanitize.sh run'
# What is instrumented: the host code of every translation unit (table packing, chunk building, buffer bookkeeping, argument checks)
# with UndefinedBehaviorSanitizer (integer overflow, shifts, null/misaligned access, array bounds of fixed-size arrays) and
# libstdc++'s container assertions (_GLIBCXX_ASSERTIONS: every std::vector index of build_chunks & co is bounds-checked).
# Device code is left alone (-fno-gpu-sanitize).
# AddressSanitizer (MODE=asan) builds, but does not run on this stack: ROCm's ASan runtime intercepts
# hsa_amd_memory_pool_allocate for device ASan, which needs an xnack+ build and an xnack-enabled driver — the first
# hipMalloc aborts with "allocator is trying to allocate 0x400000 bytes" (measured, gpurun_out/sanitize_asan.log).
# The sanitized library lives beside the product one and is selected with PQA_LIB (pyqmc_amd/_ffi.py).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODE=${MODE:-ubsan}
OUT=$ROOT/pyqmc_amd/lib/libpyqmc_amd_$MODE.so
CLANG=/opt/rocm/lib/llvm/bin/clang
if [ "$1" = build ] || [ ! -f "$OUT" ]; then
  if [ "$MODE" != ubsan ]; then echo "only MODE=ubsan is built by __graft_entry__.py --sanitize (ASan: see the note above)"; exit 1; fi
  (cd "$ROOT" && python __graft_entry__.py --sanitize)
  echo "[sanitize] built $OUT"
fi
[ "$1" = build ] && exit 0
mkdir -p "$ROOT/gpurun_out"
cd "$ROOT"
RTDIR=$(dirname $($CLANG -print-file-name=libclang_rt.ubsan_standalone-x86_64.so))
export LD_LIBRARY_PATH=$RTDIR:$LD_LIBRARY_PATH
if [ "$MODE" = asan ]; then
  export LD_PRELOAD=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
  export ASAN_OPTIONS=detect_leaks=0:detect_odr_violation=0:halt_on_error=1
fi
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 PQA_LIB=$OUT python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee "$ROOT/gpurun_out/sanitize_$MODE.log"

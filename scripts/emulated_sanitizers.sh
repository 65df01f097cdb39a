#!/bin/bash
# The product kernels + engine under AddressSanitizer / UndefinedBehaviorSanitizer: the emulated library (tests/emul) is built
# with -fsanitize=address,undefined and the emulated test module runs against it, so an out-of-bounds read of a "device"
# buffer (silent garbage on the GPU) or a signed overflow in an index computation is reported with a stack trace.
#   bash scripts/emulated_sanitizers.sh [pytest -k expression]      (-s: a sanitizer report must not be swallowed by pytest's capture)
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CXX=/opt/rocm/lib/llvm/bin/clang++
RT=$(dirname "$($CXX -print-file-name=libclang_rt.asan-x86_64.so)")
OUT=${TMPDIR:-/tmp}/hgs_simt_asan
mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-sanitize=vptr,function,alignment -fno-omit-frame-pointer -g -O1 -shared-libasan"
for f in hgs_kernels.hip hgs_engine.hip; do
  $CXX -x c++ -std=c++17 -fPIC -fno-strict-aliasing -ffp-contract=off -Wno-unknown-attributes -Wno-unused-value -Wno-pass-failed $SAN \
    -I "$ROOT/tests/emul/simt" -c "$ROOT/hdl_graph_slam_amd/csrc/$f" -o "$OUT/$f.o"
done
$CXX -x c++ -std=c++17 -fPIC $SAN -I "$ROOT/tests/emul/simt" -c "$ROOT/tests/emul/simt_runtime.cpp" -o "$OUT/rt.o"
$CXX -shared $SAN -o "$OUT/libhgs_simt.so" "$OUT/hgs_kernels.hip.o" "$OUT/hgs_engine.hip.o" "$OUT/rt.o"
cd "$ROOT"
# fibers switch stacks behind the sanitizer's back: stack-use-after-return detection off; leaks are the Python interpreter's
LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" LD_LIBRARY_PATH="$RT" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 HGS_SIMT_LIB="$OUT/libhgs_simt.so" PYTHONPATH="$ROOT/tests/emul/plugins" \
  python -m pytest tests/test_hip_parity.py tests/test_prefilter.py tests/test_map_cloud.py tests/test_odometry.py tests/test_loop_detector.py \
    tests/test_keyframe_io.py tests/test_golden.py -m gpu -p simt_everywhere -q -x -s \
    -k "${1:-not hdl32_raw and not dense and not two_engines}"   # (modules that import torch do not load under the ASan preload)

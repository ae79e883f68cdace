#!/bin/bash
# Host-side sanitizer run of the C-ABI (SURVEY.md §5 "ASAN host build of the extension"; CPU only, no GPU, no GPU sanitizer):
# csrc/api.hip's host code — weight load / fold / pack, option tables, workspaces, preprocessing descriptors, destroy — compiled as
# plain C++ with -fsanitize=address,undefined against a mock HIP runtime (tools/host_asan/hip/hip_runtime.h: device memory = malloc)
# and launcher stubs that check every range a kernel would touch (tools/host_asan/kernel_stubs.cpp), driven for all three compute dtypes
# by tools/host_asan/driver.cpp.  Exit code 0 = no report.  tests/test_host_asan.py runs this in the CPU suite.
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/rz_host_asan}
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
mkdir -p "$OUT"
python3 "$ROOT/tools/host_asan/make_weights.py" "$OUT/weights.bin"
FLAGS="-std=c++17 -g -O1 -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -I$ROOT/tools/host_asan -I$ROOT/radzero_amd/csrc -Wno-unused-function"
$CXX $FLAGS -x c++ -c "$ROOT/radzero_amd/csrc/api.hip" -o "$OUT/api.o"
$CXX $FLAGS -c "$ROOT/tools/host_asan/kernel_stubs.cpp" -o "$OUT/stubs.o"
$CXX $FLAGS -c "$ROOT/tools/host_asan/driver.cpp" -o "$OUT/driver.o"
$CXX $FLAGS "$OUT/api.o" "$OUT/stubs.o" "$OUT/driver.o" -o "$OUT/driver"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 "$OUT/driver" "$OUT/weights.bin"

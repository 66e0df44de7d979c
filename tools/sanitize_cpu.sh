#!/bin/bash
# Sanitizers on the CPU build (GPU AddressSanitizer / XNACK runs are not available on this pool).
#   1. oracle/gd4d_oracle.c with gcc -fsanitize=address,undefined under the oracle tests (golden fixtures + seeded cases)
#   2. the HOST side of libgd4d.so (argument validation of every entry point, gd4d_linear_sum_assignment_batch, launch bookkeeping)
#      with clang -fsanitize=address,undefined under the no-GPU C-ABI tests
# Two runs: gcc's and clang's ASan runtimes cannot share a process.  Writes docs/sanitizers_r06.txt.
set -u
cd "$(dirname "$0")/.."
out=docs/sanitizers_r06.txt
: > $out
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
make -s -C oracle asan || exit 1
make -s -j8 -C graph-detr4d_amd/csrc asan || exit 1
echo "== oracle/gd4d_oracle.c  (gcc $(gcc -dumpversion), -fsanitize=address,undefined): tests/test_c_oracle.py tests/test_decode_oracle.py" | tee -a $out
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" GD4D_ORACLE_SO=$PWD/oracle/_build/libgd4d_oracle_asan.so \
  python -m pytest tests/test_c_oracle.py tests/test_decode_oracle.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -4 | tee -a $out
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
echo "== libgd4d host side (hipcc --cuda-host-only -fsanitize=address,undefined): tests/test_abi.py" | tee -a $out
LD_PRELOAD="$rt" GD4D_LIB_PATH=$PWD/graph-detr4d_amd/libgd4d_asan.so \
  python -m pytest tests/test_abi.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -6 | tee -a $out
grep -c "ERROR: AddressSanitizer\|runtime error" $out | sed 's/^/sanitizer reports: /' | tee -a $out

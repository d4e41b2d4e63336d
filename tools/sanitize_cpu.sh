#!/bin/bash
# CPU sanitizer run (VERDICT r1 missing #5): the oracle and the host-side glue of the C ABI built with
# AddressSanitizer + UndefinedBehaviorSanitizer, the CPU test-suite run against them.  GPU ASan / XNACK runs are not available
# on this pool, so the device code is NOT covered; what is covered: every oracle routine the tests reach (kd-tree, heaps,
# linear algebra, ICP loops, voxel filter) and the argument / error / file-reading paths of libthreecrate_hip that run
# without a device (tc_read_kitti_bin, validation, tc_comm_* argument checks, symbol loading).
#   bash tools/sanitize_cpu.sh            -> exit code of pytest; ASan / UBSan findings abort the run
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/oracle" asan
ASAN_RT=$(gcc -print-file-name=libasan.so)
UBSAN_RT=$(gcc -print-file-name=libubsan.so)
export TC_ORACLE_LIB="$ROOT/oracle/libtc_oracle_asan.so"
# python itself is not instrumented: leak reports of the interpreter are noise, everything else is fatal
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
cd "$ROOT"
LD_PRELOAD="$ASAN_RT:$UBSAN_RT" python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
# second pass: the host side of libthreecrate_hip itself (clang's ASan runtime: it cannot share a process with gcc's, so the
# oracle-free tests only -- symbol table, error paths without a device, KITTI reader, host mirror)
if [ "${TC_SANITIZE_HIP:-1}" = "1" ]; then
    make -C "$ROOT/threecrate_amd/csrc" asan
    CLANG_ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
    unset TC_ORACLE_LIB
    TC_HIP_LIB="$ROOT/build/asan/libthreecrate_hip_asan.so" LD_PRELOAD="$CLANG_ASAN" \
        python -m pytest tests/test_abi_symbols.py tests/test_kitti_reader.py tests/test_compat_module.py -x -q -m "not gpu" -p no:cacheprovider \
        --deselect tests/test_abi_symbols.py::test_product_never_imports_the_oracle
fi

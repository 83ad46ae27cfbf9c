#!/bin/bash
# Host-side sanitizer run (CPU only, in the build container): builds libsonostft_asan.so -- AddressSanitizer and
# UndefinedBehaviorSanitizer on the host code of the C ABI, the framing and measurement stages -- and the
# lane-by-lane emulation of the kernels the same way, and runs the CPU tests that exercise them against these builds.
#   bash tools/asan_tests.sh            -> prints pytest's summary; any sanitizer report fails the run
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
python -m spectrogram_amd.build --asan > /tmp/sono_asan_build.log 2>&1 || { tail -20 /tmp/sono_asan_build.log; exit 1; }
RT="$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)"
# the emulation library: same sources the tests build with g++, here with clang and the sanitizers
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address -fsanitize=undefined -shared-libsan \
    -fno-omit-frame-pointer -o tests/host_emul/libemul_asan.so tests/host_emul/emul_generic.cpp tests/host_emul/emul_tile.cpp tests/host_emul/rowscan_check.cpp
export SONO_LIB_PATH="$ROOT/spectrogram_amd/lib/libsonostft_asan.so" SONO_EMUL_LIB="$ROOT/tests/host_emul/libemul_asan.so"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
LD_PRELOAD="$RT${LD_PRELOAD:+:$LD_PRELOAD}" python -m pytest -x -q tests/test_measure_host.py tests/test_framing_host.py tests/test_host_emul.py \
    tests/test_abi_cpu.py tests/test_tile_balance.py tests/test_wav.py tests/test_guano.py tests/test_plugin_host.py -m "not gpu" "$@"

# the Node addon (napi/sonostft_napi.c) the same way: built with clang and the sanitizers against the sanitizer build of
# the library, loaded into node with the same runtime preloaded; its CPU tests cover the export surface, argument
# marshalling and the error path of a machine without a GPU
if [ -f /usr/include/node/node_api.h ] && command -v node > /dev/null; then
    /opt/rocm/lib/llvm/bin/clang -O1 -g -fPIC -shared -std=c99 -fsanitize=address -fsanitize=undefined -shared-libsan -fno-omit-frame-pointer \
        -mllvm -asan-globals=0 -DNODE_GYP_MODULE_NAME=sonostft -DNAPI_VERSION=4 -I /usr/include/node napi/sonostft_napi.c -o napi/sonostft_asan.node \
        -L spectrogram_amd/lib -l:libsonostft_asan.so -Wl,-rpath,"$ROOT/spectrogram_amd/lib"
    # (globals are left uninstrumented: node maps the module twice, and the second registration reads as an ODR violation)
    SONO_ADDON_PATH="$ROOT/napi/sonostft_asan.node" LD_PRELOAD="$RT${LD_PRELOAD:+:$LD_PRELOAD}" python -m pytest -x -q tests/test_napi.py -m "not gpu" "$@"
fi

#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of libsfm_hip.so (device code is NOT instrumented:
# GPU ASan needs xnack+, which the pool does not offer) and the CPU test suite's ABI / host-logic tests run against
# it.  Runs in the build container (no GPU): covers library load, symbol table, argument validation and the
# no-device error paths of every entry point the tests touch.
#   bash tools/asan_host_check.sh [outdir]        -> <outdir>/libsfm_hip.so, prints the pytest summary
set -e
repo=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-/tmp/sfm_asan}
mkdir -p "$out"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -munsafe-fp-atomics -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan"
cd "$repo/structure-from-motion_amd/csrc"
pids=()
for f in sfm_core sfm_ba sfm_ba_solve sfm_ba_host sfm_ba_schur sfm_ba_schur_rows sfm_epipolar sfm_comm; do
  $HIPCC $FLAGS -c $f.hip -o "$out/$f.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan "$out"/*.o -ldl -o "$out/libsfm_hip.so"
rt=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
cd "$repo"
LD_PRELOAD="$rt" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  SFM_HIP_LIBRARY="$out/libsfm_hip.so" python -m pytest tests/test_abi_and_host.py -q -m "not gpu" -p no:cacheprovider -k "not plain_c"   # (a gcc-linked C program cannot resolve the sanitizer runtime of this build)

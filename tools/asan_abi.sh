#!/bin/bash
# Sanitizer pass over the HOST side of the C ABI: builds every translation unit of
# libcm3_hip.so with -fsanitize=address,undefined on the host code (device code objects are unchanged by the flags) into
# tools/_asan/libcm3_hip_asan.so and runs the container-side ABI tests against it -- argument validation of every entry point, the
# CopyList / TileCol / descriptor marshalling of cm3_amd/_lib.py, the error-string path -- under the sanitizer runtime.
#   bash tools/asan_abi.sh            -> prints the pytest tail and "asan: clean" / the reports; exit code 0 when clean
set -uo pipefail
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
O="${R}/tools/_asan"; mkdir -p "${O}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
SRC_ID="$(cd "${R}/cm3_amd/csrc" && { for f in $(LC_ALL=C ls *.hip *.h | LC_ALL=C sort); do cat "./${f}"; done; cat "../../include/cm3_amd.h"; } | sha256sum | cut -c1-16)"
F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -fno-slp-vectorize -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libsan -DCM3_SOURCE_ID=\"${SRC_ID}\""
pids=()
"${HIPCC}" ${F} -mllvm -amdgpu-kernarg-preload-count=16 -DCM3_PARTICLE_F32 -c "${R}/cm3_amd/csrc/particle.hip" -o "${O}/particle_f32.o" & pids+=($!)
"${HIPCC}" ${F} -DCM3_PARTICLE_F64 -c "${R}/cm3_amd/csrc/particle.hip" -o "${O}/particle_f64.o" & pids+=($!)
"${HIPCC}" ${F} -mllvm -amdgpu-kernarg-preload-count=16 -DCM3_PARTICLE_F32 -DCM3_PARTICLE_ILP_TU -c "${R}/cm3_amd/csrc/particle.hip" -o "${O}/particle_f32_ilp.o" & pids+=($!)
for f in checkers util advantage batch actor actor_checkers policy policy_checkers; do
  "${HIPCC}" ${F} -mllvm -amdgpu-kernarg-preload-count=16 -c "${R}/cm3_amd/csrc/${f}.hip" -o "${O}/${f}.o" & pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait "$p" || rc=1; done
[ $rc = 0 ] || { echo "asan build failed"; exit 2; }
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o "${O}/libcm3_hip_asan.so" "${O}"/*.o || exit 2
cd "${R}"
LOG="${O}/asan_abi.log"
LD_PRELOAD="${RT}" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 CM3_AMD_LIB="${O}/libcm3_hip_asan.so" \
  python -m pytest tests/test_abi.py -q -m "not gpu" -p no:cacheprovider > "${LOG}" 2>&1
prc=$?
tail -4 "${LOG}"
if grep -q "ERROR: AddressSanitizer\|runtime error:" "${LOG}"; then
  echo "asan: REPORTS"; grep -n "ERROR: AddressSanitizer\|runtime error:" "${LOG}" | head -20; exit 1
fi
[ $prc = 0 ] && echo "asan: clean (pytest rc 0, no AddressSanitizer / UndefinedBehaviorSanitizer report)" || { echo "asan: pytest rc ${prc}"; exit 1; }

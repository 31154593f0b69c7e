#!/bin/bash
# Round 5: builds of the WITHDRAWN transposed second layer (sources of commit c49681b, profiles/r04_policy_head.txt (9)-(12)) that
# separate the suspects: accumulator form (architectural VGPRs vs AGPRs) from co-residency (two waves per SIMD), wave priority, and
# a lane-copy divergence counter in global memory.  Only policy.o comes from the old sources (policy.hip includes actor.hip and
# particle.hip, so the object is self-contained); every other object is HEAD's (run cm3_amd/csrc/build.sh first).
# Output: cm3_amd/libcm3_hip_pf_<variant>.so (git-ignored; travels to the GPU box).  Run tools/probes/policy_fault_probe.py there.
set -eu
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"; cd "$R"
SRC_COMMIT="${SRC_COMMIT:-c49681b}"
W="$R/tools/_pf"; rm -rf "$W"; mkdir -p "$W"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
OBJ="$R/cm3_amd/csrc/_obj"
fresh() {   # fresh <dir>: the old sources
  mkdir -p "$1/csrc" "$1/include"
  for f in $(git ls-tree --name-only "$SRC_COMMIT" cm3_amd/csrc/); do git show "$SRC_COMMIT:$f" > "$1/csrc/$(basename "$f")"; done
  git show "$SRC_COMMIT:include/cm3_amd.h" > "$1/include/cm3_amd.h"
  sed -i 's#"../../include/cm3_amd.h"#"../include/cm3_amd.h"#' "$1/csrc/common.h"
}
build() {   # build <variant> <vgpr-form 0|1> [extra flags]
  local v="$1" form="$2"; shift 2
  "$HIPCC" $FLAGS -mllvm -amdgpu-mfma-vgpr-form="$form" -DCM3_SOURCE_ID="\"pf_$v\"" "$@" -Rpass-analysis=kernel-resource-usage \
      -c "$W/$v/csrc/policy.hip" -o "$W/$v/policy.o" 2> "$W/$v/resource_usage.txt"
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$R/cm3_amd/libcm3_hip_pf_$v.so" "$OBJ/particle_f32.o" "$OBJ/particle_f32_ilp.o" \
      "$OBJ/particle_f64.o" "$OBJ/checkers.o" "$OBJ/util.o" "$OBJ/advantage.o" "$OBJ/batch.o" "$OBJ/actor.o" "$OBJ/actor_checkers.o" "$W/$v/policy.o"
  python3 - "$W/$v/resource_usage.txt" "$v" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"Function Name: (\S*k_policy_rollout\S*)\n(?:.*\n){0,14}", txt):
    blk = m.group(0)
    name = m.group(1)
    g = lambda k: (re.search(k + r": (\d+)", blk) or [None, "?"])[1]
    if "Li8E" in name:       # the N = 8 builds
        print("%-12s %s  VGPRs %s AGPRs %s occupancy %s scratch %s LDS %s" % (sys.argv[2], name[-28:], g("VGPRs"), g("AGPRs"),
              g(r"Occupancy \[waves/SIMD\]"), g(r"ScratchSize \[bytes/lane\]"), g(r"LDS Size \[bytes/block\]")))
PY
}
counter_patch() {   # the lane-copy divergence counters (iv): one atomic per divergence, no printf
  python3 - "$1/csrc/policy.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
s = s.replace("namespace cm3 {\n\nstruct PolicyParams {", "namespace cm3 {\n__device__ unsigned int cm3_dbg[16];\n\nstruct PolicyParams {", 1)
s = s.replace("    V2 gl;\n    gl.x = lds.xs[rl][4]; gl.y = lds.xs[rl][5];\n    float ux = 0.0f",
              "    const V4 si_pre = si;\n    V2 gl;\n    gl.x = lds.xs[rl][4]; gl.y = lds.xs[rl][5];\n    float ux = 0.0f", 1)
probe = r'''
    {   // copies in lanes 16..63 against the row's own lane: [0..3] action, [4..7] post-step state, [8..11] pre-step state (by 16-lane row)
      const int src = lane & 15, grp = lane >> 4;
      auto ne = [&](float v) { return __float_as_uint(__shfl(v, src, 64)) != __float_as_uint(v); };
      if (__shfl(act, src, 64) != act) atomicAdd(&cm3_dbg[grp], 1u);
      if (ne(si.x) || ne(si.y) || ne(si.z) || ne(si.w)) atomicAdd(&cm3_dbg[4 + grp], 1u);
      if (ne(si_pre.x) || ne(si_pre.y) || ne(si_pre.z) || ne(si_pre.w)) atomicAdd(&cm3_dbg[8 + grp], 1u);
      if (lane == 0) atomicAdd(&cm3_dbg[15], 1u);       // ticks x waves seen (the probe is alive)
    }
'''
s = s.replace("    steps += 1;\n    ns[rl] = si;", "    steps += 1;" + probe + "    ns[rl] = si;", 1)
s = s.replace('extern "C" int cm3_policy_rollout_f32(', '''extern "C" int cm3_debug_counters(unsigned int *out, int reset) {
  unsigned int z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(cm3::cm3_dbg), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(cm3::cm3_dbg), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}

extern "C" int cm3_policy_rollout_f32(''', 1)
assert "cm3_dbg[grp]" in s and "cm3_debug_counters" in s and "si_pre = si" in s
open(p, "w").write(s)
PY
}
for v in ctrl agpr2w noprio shift9 count count_agpr2w readback; do fresh "$W/$v"; done
# (i) the two suspects apart: AGPR accumulators but forced to two waves per SIMD (<= 256 registers, spills accepted)
sed -i 's/__global__ void __launch_bounds__(256) k_policy_rollout/__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k_policy_rollout/' \
    "$W/agpr2w/csrc/policy.hip" "$W/count_agpr2w/csrc/policy.hip"
counter_patch "$W/count"
counter_patch "$W/count_agpr2w"
# (v) read the slot back right after the four copies stored it and count what differs from the lane's own value -- no shuffle
python3 - "$W/readback/csrc/policy.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
s = s.replace("namespace cm3 {\n\nstruct PolicyParams {", "namespace cm3 {\n__device__ unsigned int cm3_dbg[16];\n\nstruct PolicyParams {", 1)
s = s.replace("    ns[rl] = si;      // (one 16-byte", "    ns[rl] = si;      // (one 16-byte", 1)
s = s.replace("    wave_lds_sync();  // the other agents of this env live in the same wave\n", '''    wave_lds_sync();  // the other agents of this env live in the same wave
    {
      const V4 back = ns[rl];
      if (__float_as_uint(back.x) != __float_as_uint(si.x) || __float_as_uint(back.y) != __float_as_uint(si.y) ||
          __float_as_uint(back.z) != __float_as_uint(si.z) || __float_as_uint(back.w) != __float_as_uint(si.w))
        atomicAdd(&cm3_dbg[4 + (lane >> 4)], 1u);
      if (lane == 0) atomicAdd(&cm3_dbg[15], 1u);
    }
''', 1)
s = s.replace('extern "C" int cm3_policy_rollout_f32(', '''extern "C" int cm3_debug_counters(unsigned int *out, int reset) {
  unsigned int z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(cm3::cm3_dbg), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(cm3::cm3_dbg), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}

extern "C" int cm3_policy_rollout_f32(''', 1)
assert "back = ns[rl]" in s
open(p, "w").write(s)
PY
build ctrl 1 &
build agpr2w 0 &
build noprio 1 -DCM3_POLICY_PRIO=0 &
build shift9 1 -DCM3_POLICY_SHIFT_BIT=9 &
wait
build count 1 &
build count_agpr2w 0 &
build readback 1 &
wait
ls -la "$R"/cm3_amd/libcm3_hip_pf_*.so

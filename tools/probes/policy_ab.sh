# Same-box comparison of library builds on the whole-episode policy collection (tools/policy_row_tiles.py --worker) and, with
# MODES=1, on the launch modes / the Checkers actor.
#   LIBS="base new prio" bash tools/probes/policy_ab.sh      (base -> tools/variants/libcm3_hip_base.so, new -> the product, X -> libcm3_hip_X.so)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -n "${TESTS:-}" ] && timeout 1800 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -5
for rep in 1 2 3; do for case in "particle_stage2_antipodal 4 4096 f16x3" "particle_stage2_antipodal 4 4096 f32" "particle_merge8 8 8192 f16x3" "particle_stage2_antipodal 4 65536 f16x3"; do for b in ${LIBS:-base new}; do
  lib=""; [ $b != new ] && lib="$PWD/tools/variants/libcm3_hip_$b.so"
  echo "$case $b $(CM3_AMD_LIB=$lib timeout 300 python tools/policy_row_tiles.py --worker $case 2>&1 | tail -1 | cut -c1-26)"
done; done; done 2>&1 | tee gpurun_out/policy_ab.txt
if [ -n "${MODES:-}" ]; then for rep in 1 2; do for b in ${LIBS:-base new}; do
  lib=""; [ $b != new ] && lib="$PWD/tools/variants/libcm3_hip_$b.so"
  echo "== $b"; CM3_AMD_LIB=$lib timeout 600 python tools/policy_modes_timing.py 2>&1 | tail -8; CM3_AMD_LIB=$lib timeout 600 python tools/checkers_actor_timing.py 2>&1 | tail -12
done; done 2>&1 | tee gpurun_out/policy_modes_ab.txt; fi

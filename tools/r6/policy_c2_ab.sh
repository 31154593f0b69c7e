# same-box A/B at C2 (4096 envs x 4 agents, f16x3): product against tools/variants/libcm3_hip_${VARIANT}.so
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3 4; do
  for c in "particle_stage2_antipodal 4 4096"; do
    echo "product $c $(python tools/policy_row_tiles.py --worker $c f16x3 2>/dev/null | tail -1 | cut -c1-28)"
    echo "${VARIANT} $c $(CM3_AMD_LIB=$PWD/tools/variants/libcm3_hip_${VARIANT}.so CM3_AMD_ALLOW_STALE=1 python tools/policy_row_tiles.py --worker $c f16x3 2>/dev/null | tail -1 | cut -c1-28)"
  done
done

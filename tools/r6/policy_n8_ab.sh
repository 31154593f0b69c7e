cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3; do
  for c in "particle_merge8 8 8192" "particle_merge8 8 2048" "particle_merge8 8 65536"; do
    echo "product $c $(python tools/policy_row_tiles.py --worker $c f16x3 2>/dev/null | tail -1 | cut -c1-28)"
    echo "prev    $c $(CM3_AMD_LIB=$PWD/tools/variants/libcm3_hip_prev.so CM3_AMD_ALLOW_STALE=1 python tools/policy_row_tiles.py --worker $c f16x3 2>/dev/null | tail -1 | cut -c1-28)"
  done
done

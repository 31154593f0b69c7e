# same-box A/B of the whole-episode particle policy rollout: product against tools/variants/libcm3_hip_${VARIANT:-prev}.so, every row-tile setting
cd "${GRAFT_REPO_ROOT:-.}"
export ROW_TILES_PRECS="${ROW_TILES_PRECS:-f16x3}"
echo "== product"; python tools/policy_row_tiles.py
echo "== ${VARIANT:-prev}"; CM3_AMD_LIB=$PWD/tools/variants/libcm3_hip_${VARIANT:-prev}.so CM3_AMD_ALLOW_STALE=1 python tools/policy_row_tiles.py

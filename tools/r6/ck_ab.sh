# same-box A/B of the one-launch Checkers rollout: product against tools/variants/libcm3_hip_<name>.so (CM3_AMD_LIB; stale sources allowed)
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3 4; do
  for v in product ${VARIANTS:-base}; do
    lib=""; [ $v != product ] && lib="$PWD/tools/variants/libcm3_hip_$v.so"
    echo "$v $(CM3_AMD_LIB=$lib CM3_AMD_ALLOW_STALE=1 python tools/ck_policy_worker.py 20 2>/dev/null | tail -1 | cut -c40-)"
  done
done

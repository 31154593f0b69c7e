# same-box A/B: LDS row pad 16 halfwords (product) against 8 (tools/variants/libcm3_hip_pad8.so, built with -DCM3_CK_LD_PAD=8)
cd "${GRAFT_REPO_ROOT:-.}"
for rep in 1 2 3 4; do
  for v in product pad8; do
    lib=""; [ $v = pad8 ] && lib="$PWD/tools/variants/libcm3_hip_pad8.so"
    echo "$v $(CM3_AMD_LIB=$lib CM3_AMD_ALLOW_STALE=1 python tools/ck_policy_worker.py 20 2>/dev/null | tail -1)"
  done
done

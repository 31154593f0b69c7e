set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_actor.py tests/test_gpu_dispatch_sizes.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2 3; do for case in "particle_stage2_antipodal 4 4096 f16x3" "particle_stage2_antipodal 4 4096 f32" "particle_merge8 8 8192 f16x3" "particle_stage2_antipodal 4 65536 f16x3"; do for b in base new; do
  lib=""; [ $b = base ] && lib="$GRAFT_REPO_ROOT/cm3_amd/libcm3_hip_base.so"
  echo "$case $b $(CM3_AMD_LIB=$lib timeout 300 python tools/policy_row_tiles.py --worker $case 2>&1 | tail -1)"
done; done; done 2>&1 | tee gpurun_out/policy_ab.txt

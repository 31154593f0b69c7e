cd $GRAFT_REPO_ROOT
for E in 512 1024 2048 4096; do for prec in f16x3 f32; do for rt in 4 2 1; do
 echo "merge8 N=8 E=$E $prec RT=$rt $(CM3_POLICY_RT=$rt timeout 300 python tools/policy_row_tiles.py --worker particle_merge8 8 $E $prec 2>&1 | tail -1 | cut -c1-26)"
done; done; done

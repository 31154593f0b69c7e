# One tick per launch of the fused policy kernel (tools/probes/_bin/policy_timeline <envs> <prec> 1), per row-tile count.
cd "${GRAFT_REPO_ROOT:-.}/tools/probes/_bin"
for E in 1024 2048 4096 8192 16384; do for prec in 2 0; do for rt in 4 2 1; do
  echo "E=$E prec=$prec RT=$rt $(CM3_POLICY_RT=$rt ./policy_timeline $E $prec 1 | grep -o '[0-9.]* us per tick')"
done; done; done

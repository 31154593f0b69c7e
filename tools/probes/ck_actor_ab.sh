# Same-box comparison of library builds on the Checkers actor / policy-driven Checkers collection (tools/checkers_actor_timing.py).
#   LIBS="base new" bash tools/probes/ck_actor_ab.sh
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
[ -n "${TESTS:-}" ] && timeout 1800 python -m pytest $TESTS -x -q -m gpu 2>&1 | tail -4
for rep in 1 2 3; do for b in ${LIBS:-base new}; do
  lib=""; [ $b != new ] && lib="$PWD/tools/variants/libcm3_hip_$b.so"
  CM3_AMD_LIB=$lib timeout 600 python tools/checkers_actor_timing.py 2>/dev/null | grep f16x3 | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$b', d['what'][:14], d['envs'], d.get('us', d.get('us_per_tick')))"
done; done 2>&1 | tee gpurun_out/ck_actor_ab.txt

"""Source patches of the withdrawn-layer builds (tools/probes/policy_fault_bisect.sh): policy_fault_patch.py <variant> <csrc dir>."""
import sys

variant, d = sys.argv[1], sys.argv[2]
p = d + "/policy.hip"
s = open(p).read()


def rep(old, new, count=1):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, count)


EXPORT = '''extern "C" int cm3_debug_counters(unsigned int *out, int reset) {
  unsigned int z[32] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(cm3::cm3_dbg), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(cm3::cm3_dbg), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}

extern "C" int cm3_policy_rollout_f32('''


def counters():
    rep("namespace cm3 {\n\nstruct PolicyParams {", "namespace cm3 {\n__device__ unsigned int cm3_dbg[32];\n\nstruct PolicyParams {")
    rep('extern "C" int cm3_policy_rollout_f32(', EXPORT)


if variant.startswith("bisect"):
    counters()
    # stage hashes: h0 = positions of the other agents as read from the LDS tile, h1 = the contact forces, then the sums, the
    # velocity and the position; compared with lane l & 15 at the end of the section
    rep("    float Fx = ux * 5.0f + 0.0f, Fy = uy * 5.0f + 0.0f;\n",
        "    float Fx = ux * 5.0f + 0.0f, Fy = uy * 5.0f + 0.0f;\n    uint32_t h0 = 0, h1 = 0;\n")
    rep("      contact_force<float>(si.z - lds.xs[rj][2], si.w - lds.xs[rj][3], f_x, f_y);\n",
        "      const float pxj = lds.xs[rj][2], pyj = lds.xs[rj][3];\n"
        "      h0 = (h0 * 31u + __float_as_uint(pxj)) * 31u + __float_as_uint(pyj);\n"
        "      contact_force<float>(si.z - pxj, si.w - pyj, f_x, f_y);\n"
        "      h1 = (h1 * 31u + __float_as_uint(f_x)) * 31u + __float_as_uint(f_y);\n")
    rep("    steps += 1;\n    ns[rl] = si;", '''    steps += 1;
    {
      const int src = lane & 15, grp = lane >> 4;
      auto neu = [&](uint32_t v) { return (uint32_t)__shfl((int)v, src, 64) != v; };
      auto ne = [&](float v) { return neu(__float_as_uint(v)); };
      if (neu(h0)) atomicAdd(&cm3_dbg[0 + grp], 1u);
      if (neu(h1)) atomicAdd(&cm3_dbg[4 + grp], 1u);
      if (ne(Fx) || ne(Fy)) atomicAdd(&cm3_dbg[8 + grp], 1u);
      if (ne(si.x) || ne(si.y)) atomicAdd(&cm3_dbg[12 + grp], 1u);
      if (ne(si.z) || ne(si.w)) atomicAdd(&cm3_dbg[16 + grp], 1u);
      if (neu((uint32_t)act)) atomicAdd(&cm3_dbg[20 + grp], 1u);
      if (lane == 0) atomicAdd(&cm3_dbg[31], 1u);
    }
    ns[rl] = si;''')
elif variant == "record":
    # (E1) what a diverging copy computed: the first 16 events with lane, pair index, inputs and both results
    counters()
    rep("__device__ unsigned int cm3_dbg[32];\n", "__device__ unsigned int cm3_dbg[32];\n__device__ unsigned int cm3_rec[16][8];\n")
    rep('extern "C" int cm3_debug_counters(', '''extern "C" int cm3_debug_records(unsigned int *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cm3::cm3_rec), 16 * 8 * 4) == hipSuccess ? 0 : -1;
}
extern "C" int cm3_debug_counters(''')
    rep("      contact_force<float>(si.z - lds.xs[rj][2], si.w - lds.xs[rj][3], f_x, f_y);\n",
        """      const float dxk = si.z - lds.xs[rj][2], dyk = si.w - lds.xs[rj][3];
      contact_force<float>(dxk, dyk, f_x, f_y);
      {
        const float rx = __shfl(f_x, lane & 15, 64), ry = __shfl(f_y, lane & 15, 64);
        if (__float_as_uint(rx) != __float_as_uint(f_x) || __float_as_uint(ry) != __float_as_uint(f_y)) {
          const unsigned slot = atomicAdd(&cm3_dbg[0], 1u);
          if (slot < 16) {
            cm3_rec[slot][0] = (unsigned)lane | ((unsigned)k << 8) | ((unsigned)t << 16) | ((unsigned)w << 24);
            cm3_rec[slot][1] = __float_as_uint(dxk); cm3_rec[slot][2] = __float_as_uint(dyk);
            cm3_rec[slot][3] = __float_as_uint(f_x); cm3_rec[slot][4] = __float_as_uint(f_y);
            cm3_rec[slot][5] = __float_as_uint(rx); cm3_rec[slot][6] = __float_as_uint(ry);
            cm3_rec[slot][7] = (unsigned)blockIdx.x;
          }
        }
      }
""")
elif variant in ("nops_after_head", "nops_before_head", "waitall_after_head", "nops_after_chain0_inputs"):
    pad = "    __builtin_amdgcn_sched_barrier(0);\n" + "    asm volatile(\"s_nop 15\");\n" * 16 + "    __builtin_amdgcn_sched_barrier(0);\n"
    if variant == "nops_after_head":          # 256 wait states between the action broadcast and the physics
        rep("    const int act = bcast_row0(actor_pick(pr, u));\n", "    const int act = bcast_row0(actor_pick(pr, u));\n" + pad)
    elif variant == "nops_before_head":       # ... between the barrier behind the second layer and the head
        rep("    __builtin_amdgcn_s_setprio(CM3_POLICY_PRIO);\n    float pr[kA];\n", "    __builtin_amdgcn_s_setprio(CM3_POLICY_PRIO);\n" + pad + "    float pr[kA];\n")
    elif variant == "waitall_after_head":     # every outstanding memory / LDS operation retired before the physics
        rep("    const int act = bcast_row0(actor_pick(pr, u));\n",
            "    const int act = bcast_row0(actor_pick(pr, u));\n    __builtin_amdgcn_sched_barrier(0);\n    __builtin_amdgcn_s_waitcnt(0);\n    __builtin_amdgcn_sched_barrier(0);\n")
    else:                                     # the row's own state and goal are read first, then 256 wait states, then the chains
        rep("    float ux = 0.0f, uy = 0.0f;\n    if (act == 1) ux = -1.0f;", pad + "    float ux = 0.0f, uy = 0.0f;\n    if (act == 1) ux = -1.0f;")
elif variant == "sgpr100":
    # the force constant from a scalar register instead of a 32-bit literal in the instruction (chain 0's x path is the only
    # place where the compiler multiplies by the literal 100.0)
    q = open(d + "/particle.hip").read()
    old = "  const R kMargin = R(1e-3), kForce = R(1e+2), kDistMin = R(0.15) + R(0.15);\n  const R dist = Math<R>::sqrt(d2);"
    assert old in q
    q = q.replace(old, "  const R kMargin = R(1e-3), kDistMin = R(0.15) + R(0.15);\n"
                       "  const R kForce = (R)__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x42c80000));\n  const R dist = Math<R>::sqrt(d2);", 1)
    open(d + "/particle.hip", "w").write(q)
elif variant == "record2":
    # (E2) the intermediates of a diverging contact force: quotients 100 dx / dist, 100 dy / dist and the penetration term
    counters()
    rep("__device__ unsigned int cm3_dbg[32];\n", "__device__ unsigned int cm3_dbg[32];\n__device__ unsigned int cm3_rec[16][8];\n__device__ float cm3_q[3];\n")
    rep('extern "C" int cm3_debug_counters(', '''extern "C" int cm3_debug_records(unsigned int *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cm3::cm3_rec), 16 * 8 * 4) == hipSuccess ? 0 : -1;
}
extern "C" int cm3_debug_counters(''')
    q = open(d + "/particle.hip").read()
    old = "  f_x = kForce * dx / dist * pen;\n  f_y = kForce * dy / dist * pen;\n"
    assert old in q
    q = q.replace(old, "  const R qx = kForce * dx / dist, qy = kForce * dy / dist;\n  f_x = qx * pen;\n  f_y = qy * pen;\n"
                       "  if (dbg) { dbg[0] = (float)qx; dbg[1] = (float)qy; dbg[2] = (float)pen; }\n", 1)
    q = q.replace("template <typename R> __device__ __forceinline__ void contact_force_near(R dx, R dy, R d2, R &f_x, R &f_y) {",
                  "template <typename R> __device__ __forceinline__ void contact_force_near(R dx, R dy, R d2, R &f_x, R &f_y, float *dbg = nullptr) {", 1)
    q = q.replace("template <typename R> __device__ __forceinline__ void contact_force(R dx, R dy, R &f_x, R &f_y) {",
                  "template <typename R> __device__ __forceinline__ void contact_force(R dx, R dy, R &f_x, R &f_y, float *dbg = nullptr) {", 1)
    q = q.replace("  if (!(d2 >= Thresh<R>::kSkip2)) contact_force_near<R>(dx, dy, d2, f_x, f_y);", "  if (!(d2 >= Thresh<R>::kSkip2)) contact_force_near<R>(dx, dy, d2, f_x, f_y, dbg);", 1)
    open(d + "/particle.hip", "w").write(q)
    rep("      contact_force<float>(si.z - lds.xs[rj][2], si.w - lds.xs[rj][3], f_x, f_y);\n",
        """      float dbgv[3] = {0.0f, 0.0f, 0.0f};
      contact_force<float>(si.z - lds.xs[rj][2], si.w - lds.xs[rj][3], f_x, f_y, dbgv);
      {
        const float rx = __shfl(f_x, lane & 15, 64), ry = __shfl(f_y, lane & 15, 64);
        if (__float_as_uint(rx) != __float_as_uint(f_x) || __float_as_uint(ry) != __float_as_uint(f_y)) {
          const unsigned slot = atomicAdd(&cm3_dbg[0], 1u);
          if (slot < 16) {
            cm3_rec[slot][0] = (unsigned)lane | ((unsigned)k << 8) | ((unsigned)t << 16) | ((unsigned)w << 24);
            cm3_rec[slot][1] = __float_as_uint(dbgv[0]); cm3_rec[slot][2] = __float_as_uint(dbgv[1]);
            cm3_rec[slot][3] = __float_as_uint(f_x); cm3_rec[slot][4] = __float_as_uint(f_y);
            cm3_rec[slot][5] = __float_as_uint(rx); cm3_rec[slot][6] = __float_as_uint(ry);
            cm3_rec[slot][7] = __float_as_uint(dbgv[2]);
          }
        }
      }
""")
elif variant == "final_mul_scalar":
    # only the LAST two multiplies of the contact chain as scalar v_mul_f32 (every other packed instruction stays)
    q = open(d + "/particle.hip").read()
    old = "  f_x = kForce * dx / dist * pen;\n  f_y = kForce * dy / dist * pen;\n"
    assert old in q
    q = q.replace(old, "  const R qx = kForce * dx / dist, qy = kForce * dy / dist;\n"
                       "  if constexpr (sizeof(R) == 4) {\n"
                       "    asm(\"v_mul_f32 %0, %1, %2\" : \"=v\"(f_x) : \"v\"(qx), \"v\"(pen));\n"
                       "    asm(\"v_mul_f32 %0, %1, %2\" : \"=v\"(f_y) : \"v\"(qy), \"v\"(pen));\n"
                       "  } else {\n    f_x = qx * pen;\n    f_y = qy * pen;\n  }\n", 1)
    open(d + "/particle.hip", "w").write(q)
elif variant.startswith("cut"):
    # delta debugging of the failing kernel: the lane-copy counter stays, parts of the kernel go (round 5, towards a smaller reproducer)
    counters()
    rep("    steps += 1;\n    ns[rl] = si;", '''    steps += 1;
    {
      const int src = lane & 15, grp = lane >> 4;
      auto ne = [&](float v) { return __float_as_uint(__shfl(v, src, 64)) != __float_as_uint(v); };
      if (ne(si.x) || ne(si.y) || ne(si.z) || ne(si.w)) atomicAdd(&cm3_dbg[4 + grp], 1u);
      if (lane == 0) atomicAdd(&cm3_dbg[31], 1u);
    }
    ns[rl] = si;''')
    if "A" in variant:      # the instruction form forced (every chain ends in v_pk_mul_f32 ... op_sel:[0,1]) so that cuts cannot lose it
        q = open(d + "/particle.hip").read()
        old = "  f_x = kForce * dx / dist * pen;\n  f_y = kForce * dy / dist * pen;\n"
        assert old in q
        q = q.replace(old, """  if constexpr (sizeof(R) == 4) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v qv = {kForce * dx / dist, kForce * dy / dist}, pv = {dist, pen}, fv;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(fv) : "v"(qv), "v"(pv));
    f_x = fv[0];
    f_y = fv[1];
  } else {
    f_x = kForce * dx / dist * pen;
    f_y = kForce * dy / dist * pen;
  }
""", 1)
        open(d + "/particle.hip", "w").write(q)
    if "1" in variant:      # cut1: only the first contact pair is evaluated
        rep("    for (int k = 0; k < N - 1; ++k) {  // the reference's accumulation order", "    for (int k = 0; k < 1; ++k) {  // the reference's accumulation order")
    if "2" in variant:      # cut2: no rewards / collisions / resets / trajectory stores behind the exchange: straight to the next tick's tile
        a = s.index("    CM3_STAMP(8, false);\n    // ---- reward / reached / collisions")
        b = s.index("    CM3_STAMP(10, false);\n    // ---- trajectory stores + the LDS tile of the next tick")
        s_ = s[:a] + "    float rew = 0.0f; bool was_reset = false; float pr_dummy = pr[0]; (void)pr_dummy;\n" + s[b:]
        globals()["s"] = s_
    if "3" in variant:      # cut3: the head's result is not used: action 1 for everyone (the head still runs)
        rep("    const int act = bcast_row0(actor_pick(pr, u));", "    const int act = 1 + 0 * bcast_row0(actor_pick(pr, u));")
elif variant.startswith("agg_"):
    # WHICH part of the other wave's second layer does the fault need?  The lane-copy counters are the detector (they do not depend on
    # the layer computing the right thing); the layer is changed:
    #   agg_nomfma   no matrix instruction in it (the LDS reloads stay, their values are summed with vector adds)
    #   agg_nolds    the matrix instructions stay, their B operands are loaded ONCE (s = 0) instead of reloaded every k-step
    #   agg_swap     operands the other way round (activations as A, weights as B): the order of the clean build, everything else as it is
    #   agg_one      ONE product per k-step instead of three (a third of the matrix instructions)
    counters()
    rep("    steps += 1;\n    ns[rl] = si;", '''    steps += 1;
    {
      const int src = lane & 15, grp = lane >> 4;
      auto ne = [&](float v) { return __float_as_uint(__shfl(v, src, 64)) != __float_as_uint(v); };
      if (ne(si.x) || ne(si.y) || ne(si.z) || ne(si.w)) atomicAdd(&cm3_dbg[4 + grp], 1u);
      if (lane == 0) atomicAdd(&cm3_dbg[31], 1u);
    }
    ns[rl] = si;''')
    q = open(d + "/actor.hip").read()
    old = """      for (int t = 0; t < RT; ++t) accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.bwh[s], al[t], accs[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.bwh[s], ah[t], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t) accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.bwl[s], ah[t], accs[t], 0, 0, 0);
"""
    assert old in q
    if variant == "agg_nomfma":
        new = """      for (int t = 0; t < RT; ++t) {
        acc[t][0] += (float)ah[t][0] + (float)al[t][1]; acc[t][1] += (float)ah[t][2] + (float)al[t][3];
        acc[t][2] += (float)ah[t][4] + (float)al[t][5]; acc[t][3] += (float)ah[t][6] + (float)al[t][7];
        accs[t][0] += (float)b.bwh[s][0]; accs[t][1] += (float)b.bwl[s][1];
      }
"""
        q = q.replace(old, new)
    elif variant == "agg_swap":
        q = q.replace(old, old.replace("(b.bwh[s], al[t],", "(al[t], b.bwh[s],").replace("(b.bwh[s], ah[t],", "(ah[t], b.bwh[s],").replace("(b.bwl[s], ah[t],", "(ah[t], b.bwl[s],"))
    elif variant == "agg_one":
        q = q.replace(old, """      for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b.bwh[s], ah[t], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t) accs[t][0] += (float)al[t][0] + (float)b.bwl[s][0];
""")
    elif variant == "agg_nolds":
        ld = """      f16x8 ah[RT], al[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        ah[t] = *reinterpret_cast<const f16x8 *>(&lds.h1h[16 * t + col][32 * s + 8 * hi]);
        al[t] = *reinterpret_cast<const f16x8 *>(&lds.h1l[16 * t + col][32 * s + 8 * hi]);
      }
"""
        assert ld in q
        q = q.replace(ld, ld.replace("[32 * s + 8 * hi]", "[8 * hi]"))      # the same address every k-step: the compiler hoists the loads
    else:
        raise SystemExit("unknown variant " + variant)
    open(d + "/actor.hip", "w").write(q)
elif variant in ("ctrl2", "nopk", "noslp"):
    pass
else:
    raise SystemExit("unknown variant " + variant)
# the old sources predate ABI 6: the entry point the current binding requires, as a stub
if "cm3_policy_force_row_tiles" not in s:
    s += '\nextern "C" int cm3_policy_force_row_tiles(int32_t) { return 0; }\n'
open(p, "w").write(s)

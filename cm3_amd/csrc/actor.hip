// On-device actor for the particle env: forward pass + epsilon-mixed categorical sampling, so that
// policy-driven rollouts never leave the GPU.
//
// Reference being replaced:
//   networks.actor_particle        /root/reference/alg/networks.py:517-538
//   probs = (1-eps) probs + eps/5  /root/reference/alg/alg_credit.py:119
//   action ~ multinomial(log p)    /root/reference/alg/alg_credit.py:120, run_actor :249-270
// One row per agent of every env (row r = e*N + i, the order of obs_others [E][N][L]).
//
// Mapping: a 256-thread workgroup (4 waves, one per SIMD of a CU) owns 64 rows.  The 192 -> 64 second layer (87 % of
// the FLOPs) runs on the matrix cores with the exact-f32 MFMA (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, so
// the numerics are those of a plain f32 loop); wave w owns columns [16w, 16w+16).  The first layers are evaluated
// directly in the MFMA A-operand layout (details at the kernel), the four column quarters meet in LDS, and wave 0
// finishes with the 5-way output layer, softmax, epsilon mixing and inverse-CDF sampling from the build's Philox
// stream.  float32 throughout, like the reference graph.  (A first version streamed the weights through the scalar
// cache into SGPR operands of VALU FMAs: correct, but SGPR capacity made it latency-bound -- 23 us per tick at
// 4096 envs x 4 agents.)
#include "actor_common.h"

namespace cm3 {

constexpr int kH1S = 64, kH1O = 128, kH2 = 64;
constexpr int kH2W = kH2 + 1;   // h2 row in LDS (odd stride)

struct ActorParams {
  int E, stage;
  float eps;
  int bf16;  // precision of the second layer: kPrecF32 / kPrecBf16 / kPrecF16x3 (see ActorGeom)
  int64_t env_id_base;
  uint64_t seed;
  const float *obs_others, *state, *goals;
  const int32_t *meta, *episode;
  int32_t *actions;
  float *probs;
  const float *eps_dev;  // optional: epsilon read from the device at launch (graph replays follow the annealing)
  const float *w_self, *b_self, *w_self_h2, *w_oth, *b_oth, *w_oth_h2, *b_h2, *w_out, *b_out;  // pack kernel only
  const float *packed;  // kernel-layout weights (PackLayout), written by k_actor_pack
};

// Kernel-layout weight buffer (floats), produced once per weight update by cm3_actor_particle_pack so that the forward
// kernel fills its LDS tables with straight 16-byte copies and loads its W2 slice with 12 (f32) or 6 (bf16) vector loads
// per lane instead of ~65 scalar loads and a bank-conflicted in-kernel transpose (8k cycles -> see DESIGN.md).
template <int N> struct PackLayout {
  static constexpr int L = 4 * (N > 1 ? N - 1 : 1);
  static constexpr int SW = 8, OW = L + 4, KU = kH1S + kH1O;
  static constexpr int kSelf = 0;                       // [64][8]   unit-major: 6 weights, bias, pad
  static constexpr int kOth = kSelf + kH1S * SW;         // [128][OW] unit-major: L weights, bias, pad
  static constexpr int kOut = kOth + kH1O * OW;          // [64*5] w_out, [5] b_out, pad to 328
  static constexpr int kOutSize = 328;
  static constexpr int kTables = kOut + kOutSize;        // everything above is copied to LDS verbatim
  static constexpr int kBh2 = kTables;                   // [64]
  static constexpr int kW2f = kBh2 + kH2;                // [4 waves][64 lanes][48 k-steps] f32 B operands
  static constexpr int kW2b = kW2f + 4 * 64 * (KU / 4);  // [4][64][6][8] bf16 B operands (stored in float slots)
  static constexpr int kW2x = kW2b + 4 * 64 * (KU / 4) / 2;  // [4][64][6][hi | lo][8] split-float16 B operands (kPrecF16x3)
  static constexpr int kTotal = kW2x + 4 * 64 * (KU / 4);
};

// Split float16 (kPrecF16x3): x = hi + lo with hi = f16(x) and the residual stored SCALED, lo' = f16((x - hi) * 2^11).  |x - hi| is at
// most 2^-11 |x|, so lo' lives in hi's own exponent range: without the scale the residual of every |x| < 0.125 fell into float16's
// subnormals (absolute error ~3e-8 per factor whatever x is -- ADVICE r3) and small weights against large activations could leave
// the 2e-5 bound.  The two small products are accumulated apart from hi x hi and folded in with one exact multiply by 2^-11.
constexpr float kLoScale = 2048.0f, kLoUnscale = 1.0f / 2048.0f;

template <int N> __global__ void __launch_bounds__(256) k_actor_pack(const ActorParams p, float *out) {
  using PL = PackLayout<N>;
  const bool stage2 = p.stage > 1;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < PL::kTotal; t += gridDim.x * 256) {
    float v = 0.0f;
    if (t < PL::kOth) {
      const int k = t / PL::SW, in = t - k * PL::SW;
      v = in < 6 ? p.w_self[in * kH1S + k] : (in == 6 ? p.b_self[k] : 0.0f);
    } else if (t < PL::kOut) {
      const int u = t - PL::kOth, k = u / PL::OW, in = u - k * PL::OW;
      if (stage2) v = in < PL::L ? p.w_oth[in * kH1O + k] : (in == PL::L ? p.b_oth[k] : 0.0f);
    } else if (t < PL::kTables) {
      const int u = t - PL::kOut;
      v = u < kH2 * kA ? p.w_out[u] : (u < kH2 * kA + kA ? p.b_out[u - kH2 * kA] : 0.0f);
    } else if (t < PL::kW2f) {
      v = p.b_h2[t - PL::kBh2];
    } else if (t < PL::kW2b) {
      const int u = t - PL::kW2f, s = u % (PL::KU / 4), lane = (u / (PL::KU / 4)) & 63, w = u / (PL::KU / 4) / 64;
      const int k = 4 * s + (lane >> 4), j = 16 * w + (lane & 15);  // B[k = l>>4][j = l&15] of k-step s
      v = k < kH1S ? p.w_self_h2[k * kH2 + j] : (stage2 ? p.w_oth_h2[(k - kH1S) * kH2 + j] : 0.0f);
    } else if (t < PL::kW2x) {
      // two bf16 per float slot: element index eidx = 2 (t - kW2b) + {0,1} in [w][lane][s(6)][q(8)]
      __bf16 pair[2];
      for (int h = 0; h < 2; ++h) {
        const int eidx = 2 * (t - PL::kW2b) + h;
        const int q = eidx & 7, s = (eidx >> 3) % (PL::KU / 32), lane = ((eidx >> 3) / (PL::KU / 32)) & 63;
        const int w = (eidx >> 3) / (PL::KU / 32) / 64;
        const int k = 32 * s + 8 * (lane >> 4) + q, j = 16 * w + (lane & 15);
        const float wv = k < kH1S ? p.w_self_h2[k * kH2 + j] : (stage2 ? p.w_oth_h2[(k - kH1S) * kH2 + j] : 0.0f);
        pair[h] = (__bf16)wv;
      }
      __builtin_memcpy(&v, pair, 4);
    } else {
      // split float16 (kPrecF16x3): eidx = 2 (t - kW2x) + {0,1} in [w][lane][s(6)][part(2: hi, lo)][q(8)];
      // hi = f16(w), lo = f16(w - hi): hi + lo carries 22 of the weight's 24 significand bits
      _Float16 pair[2];
      for (int h = 0; h < 2; ++h) {
        const int eidx = 2 * (t - PL::kW2x) + h;
        const int q = eidx & 7, part = (eidx >> 3) & 1, s = (eidx >> 4) % (PL::KU / 32), lane = ((eidx >> 4) / (PL::KU / 32)) & 63;
        const int w = (eidx >> 4) / (PL::KU / 32) / 64;
        const int k = 32 * s + 8 * (lane >> 4) + q, j = 16 * w + (lane & 15);
        const float wv = k < kH1S ? p.w_self_h2[k * kH2 + j] : (stage2 ? p.w_oth_h2[(k - kH1S) * kH2 + j] : 0.0f);
        const _Float16 hi = (_Float16)wv;
        pair[h] = part == 0 ? hi : (_Float16)((wv - (float)hi) * kLoScale);   // (the residual is kept scaled: see kLoScale)
      }
      __builtin_memcpy(&v, pair, 4);
    }
    out[t] = v;
  }
}


// ---- building blocks shared by k_actor_particle (one tick) and k_policy_rollout (policy.hip: a whole policy-driven
// episode in one launch).  A workgroup = 4 waves (one per SIMD of a CU) = 64 agent rows (row r = e*N + i).
//   Phase A (matrix cores): wave w evaluates a quarter of the first-layer units -- 16 of branch_self and 32 of
//     actor_others -- for all 64 rows (K = 6 -> 8 and K = L); bias + relu on the C tile; results go to LDS h1[row][unit]
//     (unit = k of layer 2).  A[i = l&15][k = l>>4] = input k of row 16t + (l&15) from the xs tile,
//     B[k = l>>4][j = l&15] = weight of input k for unit j from the unit-major LDS tables.
//     (A VALU version with one row per lane and broadcast weight reads took 7.0k cycles; this one 3.7k.)
//   Phase B (matrix cores): wave w owns columns [16w, 16w+16) of the 192 -> 64 second layer,
//     v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fma chain), 4 row tiles x 48 k-steps:
//       A[i = l&15][k = l>>4]  = h1s[16t + (l&15)][4s + (l>>4)]        one ds_read_b32 per MFMA
//       B[k = l>>4][j = l&15]  = W2[4s + (l>>4)][16w + (l&15)]         48 VGPRs per lane, loaded once
//       C  col = l&15, row = 4 (l>>4) + reg  -> relu(C + b) to LDS h2s; each wave then finishes 16 whole rows.
//   BF16 == true (opt-in, cm3_actor_particle_desc.precision = 1): the first-layer activations are stored as bf16 and the
//     second layer runs on v_mfma_f32_16x16x32_bf16 (f32 accumulate): 6 k-steps instead of 48, 16x the MFMA rate.
//       A[i = l&15][k = 8 (l>>4) + q] = h1b[16t + (l&15)][32s + 8 (l>>4) + q],  q = 0..7   one ds_read_b128 per MFMA
//       B[k = 8 (l>>4) + q][j = l&15] = bf16(W2[32s + 8 (l>>4) + q][16w + (l&15)])          24 VGPRs per lane
//     Activations and W2 are rounded to bf16 (relative 2^-9): probabilities move by up to ~1e-2, so this is NOT the
//     parity path; everything else (first layers, output layer, softmax, sampling) stays float32.
//   History: weights streamed through SGPRs into VALU FMAs: 23 us per tick at 4096 envs x 4 agents; first layer
//   evaluated directly in the A-operand layout in all four waves: 2750 VALU instructions per wave, 14.5 us (PMC runs in
//   profiles/); this structure: 9.5 us.
// precision of the 192 -> 64 second layer (cm3_actor_particle_desc.precision)
//   kPrecF32     exact float32 MFMA (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain), 192 MFMAs of 32 cycles per wave
//   kPrecBf16    activations and weights rounded to bf16, f32 accumulate: 24 MFMAs; probabilities move by ~1e-2 -- not a parity path
//   kPrecF16x3   (round 3) SPLIT float16: x = hi + lo with hi = f16(x), lo = f16(x - hi), for activations and weights alike, and
//                x w ~ hi hi + hi lo + lo hi on v_mfma_f32_16x16x32_f16 (products exact, f32 accumulate): 72 MFMAs of ~8 cycles.
//                float16 keeps 11 significand bits, so hi + lo keeps 22 of float32's 24; what is dropped is lo lo and the second
//                residuals, ~3 x 2^-22 relative per product.  Measured against the float64 oracle with deliberately large
//                weights (tests/test_gpu_actor.py) the probabilities stay inside the 2e-5 the float32 path is held to.
//                (The same split in bf16 -- 16 bits per factor -- was tried first: 3.7e-5 .. 1.1e-4, outside the bound;
//                profiles/r03_actor_split_precision.txt.)  Range: float16 overflows at 65504; a first-layer activation beyond
//                that (weights ~3 orders of magnitude above anything a trained policy holds) saturates instead of matching.
constexpr int kPrecF32 = 0, kPrecBf16 = 1, kPrecF16x3 = 2;


// RT = 16-row tiles per workgroup (4: 64 agent rows, the one-tick actor kernel; the fused policy rollout also runs 2 or 1 so that a
// small batch still puts several workgroups on every CU, see policy.hip)
template <int N, int PREC, int RT = 4> struct ActorGeom {
  static constexpr int ROWS = 16 * RT;
  static constexpr bool BF16 = PREC != kPrecF32;     // activations stored as bf16 (one plane, or hi + lo planes)
  static constexpr int PLANES = PREC == kPrecF16x3 ? 2 : 1;
  static constexpr int L = 4 * (N > 1 ? N - 1 : 1);
  static constexpr int SW = 8;            // ws_self row: 6 weights, bias, pad
  static constexpr int OW = L + 4;        // ws_oth row: L weights, bias, pad (multiple of 4 floats)
  static constexpr int KU = kH1S + kH1O;  // 192 first-layer units = K of the second layer
  static constexpr int HB = KU + 8;       // bf16 row: 200 halfwords = 400 B (16-byte aligned rows)
  static constexpr int XW = 6 + L + 1;    // input tile row: [v_obs(4) | v_goal(2) | obs_others(L)], odd stride
  static constexpr int HS = KU + 2;        // f32 row: 194 floats (8-byte aligned rows; reads of 16 rows x 2 k hit 32 distinct banks)
  static constexpr int kH1Floats = BF16 ? PLANES * (ROWS * HB) / 2 : ROWS * HS;
  static_assert(ROWS * kH2W <= kH1Floats, "h2 must fit into the h1 storage");
  // one- and two-tile workgroups (the fused policy rollout on small batches) keep h2 in LDS of its own: the barrier between the
  // second layer's last read of h1 and the first store of h2 goes away, one of the tick's four (+8 KB at two tiles; several such
  // workgroups still share a CU).  Four-tile workgroups stay at two per CU by reusing the h1 storage.
  static constexpr bool H2SEP = RT <= 2;
  static constexpr int kH2Floats = H2SEP ? ROWS * kH2W : 4;
};

// Views into the workgroup's LDS (declared by the kernel with CM3_ACTOR_LDS).  h2 reuses the h1 storage once every wave
// is done reading h1, which keeps the workgroup at 66 KB (f32) / 42 KB (bf16) so that two workgroups fit a CU.
template <int N, int PREC, int RT = 4> struct ActorLds {
  using G = ActorGeom<N, PREC, RT>;
  float (*ws_self)[G::SW];
  float (*ws_oth)[G::OW];
  const float *wout;
  float (*h1s)[G::HS];
  __bf16 (*h1b)[G::HB];     // kPrecBf16: bf16 activations
  _Float16 (*h1h)[G::HB];   // kPrecF16x3: the hi plane ...
  _Float16 (*h1l)[G::HB];   // ... and the lo plane
  float (*h2s)[kH2W];
  float (*xs)[G::XW];
  float *tables;
};

#define CM3_ACTOR_LDS_RT(N_, BF16_, RT_, name)                                                                 \
  __shared__ __attribute__((aligned(16))) float name##_tables[ActorTabRegs<N_>::kPadded];                      \
  __shared__ __attribute__((aligned(16))) float name##_h1raw[ActorGeom<N_, BF16_, RT_>::kH1Floats];            \
  __shared__ float name##_xs[16 * RT_][ActorGeom<N_, BF16_, RT_>::XW];                                         \
  __shared__ __attribute__((aligned(16))) float name##_h2raw[ActorGeom<N_, BF16_, RT_>::kH2Floats];                                         \
  ActorLds<N_, BF16_, RT_> name;                                                                               \
  name.tables = name##_tables;                                                                                 \
  name.ws_self = reinterpret_cast<float (*)[ActorGeom<N_, BF16_>::SW]>(&name##_tables[PackLayout<N_>::kSelf]); \
  name.ws_oth = reinterpret_cast<float (*)[ActorGeom<N_, BF16_>::OW]>(&name##_tables[PackLayout<N_>::kOth]);   \
  name.wout = &name##_tables[PackLayout<N_>::kOut];                                                            \
  name.h1s = reinterpret_cast<float (*)[ActorGeom<N_, BF16_>::HS]>(name##_h1raw);                              \
  name.h1b = reinterpret_cast<__bf16 (*)[ActorGeom<N_, BF16_>::HB]>(name##_h1raw);                             \
  name.h1h = reinterpret_cast<_Float16 (*)[ActorGeom<N_, BF16_>::HB]>(name##_h1raw);                           \
  name.h1l = reinterpret_cast<_Float16 (*)[ActorGeom<N_, BF16_>::HB]>(name##_h1raw) + 16 * RT_;                \
  name.h2s = reinterpret_cast<float (*)[kH2W]>(ActorGeom<N_, BF16_, RT_>::H2SEP ? name##_h2raw : name##_h1raw); \
  name.xs = name##_xs
#define CM3_ACTOR_LDS(N_, BF16_, name) CM3_ACTOR_LDS_RT(N_, BF16_, 4, name)

// first-layer tables + output layer -> LDS: straight 16-byte copies of the packed prefix, in two halves -- every request of the
// kernel's entry is issued before the first wait (round 4: the copy was a loop of load / wait / store, 2.8 trips at N = 4 = three
// memory round trips in a row before the weight slice and the input rows were even requested; 6 000 - 7 000 cycles from the
// first instruction to the barrier, tools/probes/policy_timeline.hip)
template <int N> struct ActorTabRegs {
  static constexpr int kT4 = PackLayout<N>::kTables / 4;
  static constexpr int NV = (kT4 + 255) / 256;
  static constexpr int kPadded = NV * 256 * 4;   // floats of the LDS copy: whole trips of 256 lanes x 16 bytes (see actor_tables_store)
  f32x4 v[NV];
};
template <int N> __device__ __forceinline__ void actor_tables_fetch(const float *packed, int tid, ActorTabRegs<N> &r) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(packed);
#pragma unroll
  for (int k = 0; k < ActorTabRegs<N>::NV; ++k) {
    const int idx = tid + 256 * k;   // (the last trip reads a clamped index instead of branching: straight-line requests)
    r.v[k] = src[256 * (k + 1) <= ActorTabRegs<N>::kT4 ? idx : (idx < ActorTabRegs<N>::kT4 ? idx : ActorTabRegs<N>::kT4 - 1)];
  }
}
template <int N, int PREC, int RT>
__device__ __forceinline__ void actor_tables_store(const ActorLds<N, PREC, RT> &lds, int tid, const ActorTabRegs<N> &r) {
  f32x4 *dst = reinterpret_cast<f32x4 *>(lds.tables);
#pragma unroll
  for (int k = 0; k < ActorTabRegs<N>::NV; ++k) {
    // (unconditional: the LDS array is padded to whole trips and the lanes past the end store their clamped copy into the padding.
    // With the store under a branch the compiler sank the last request into that branch -- a second round trip behind a
    // wait for everything)
    dst[tid + 256 * k] = r.v[k];
  }
}

// B operands of this wave's 16 columns of W2 = [W_branch_self_h2 ; W_others_h2] and their bias, kept in VGPRs
template <int N, int PREC> struct ActorB {
  using G = ActorGeom<N, PREC>;
  float bw[PREC == kPrecF32 ? G::KU / 4 : 1];
  bf16x8 bwb[PREC == kPrecBf16 ? G::KU / 32 : 1];       // kPrecBf16: bf16 weights
  f16x8 bwh[PREC == kPrecF16x3 ? G::KU / 32 : 1];       // kPrecF16x3: the hi parts ...
  f16x8 bwl[PREC == kPrecF16x3 ? G::KU / 32 : 1];       // ... and the lo parts
  float bias_h2;
};

template <int N, int PREC>
__device__ __forceinline__ void actor_load_b(const float *packed, int w, int lane, ActorB<N, PREC> &b) {
  using G = ActorGeom<N, PREC>;
  using PL = PackLayout<N>;
  if constexpr (PREC == kPrecF16x3) {
    const uint4 *src = reinterpret_cast<const uint4 *>(packed + PL::kW2x) + (size_t)(w * 64 + lane) * (G::KU / 16);
#pragma unroll
    for (int s = 0; s < G::KU / 32; ++s) {
      const uint4 hi = src[2 * s], lo = src[2 * s + 1];
      __builtin_memcpy(&b.bwh[s], &hi, 16);
      __builtin_memcpy(&b.bwl[s], &lo, 16);
    }
  } else if constexpr (PREC == kPrecBf16) {
    const uint4 *src = reinterpret_cast<const uint4 *>(packed + PL::kW2b) + (size_t)(w * 64 + lane) * (G::KU / 32);
#pragma unroll
    for (int s = 0; s < G::KU / 32; ++s) {
      const uint4 v = src[s];
      __builtin_memcpy(&b.bwb[s], &v, 16);
    }
  } else {
    const float4 *src = reinterpret_cast<const float4 *>(packed + PL::kW2f) + (size_t)(w * 64 + lane) * (G::KU / 16);
#pragma unroll
    for (int s4 = 0; s4 < G::KU / 16; ++s4) {
      const float4 v = src[s4];
      b.bw[4 * s4 + 0] = v.x; b.bw[4 * s4 + 1] = v.y; b.bw[4 * s4 + 2] = v.z; b.bw[4 * s4 + 3] = v.w;
    }
  }
  b.bias_h2 = packed[PL::kBh2 + 16 * w + (lane & 15)];
}

// Phase-A B operands of one lane: unit 16w + col of branch_self and units 32w + 16cq + col of actor_others, k = 4s + hi
// (B[k = l>>4][j = l&15]), read from the unit-major first-layer tables in LDS.
#ifndef CM3_F16OTH_MIN
#define CM3_F16OTH_MIN 16
#endif
template <int N> struct ActorFirstB {
  static constexpr int L4 = (4 * (N > 1 ? N - 1 : 1)) / 4;
  // kPrecF16x3 with 16 <= L <= 32 inputs (N = 8: 28): actor_others runs in split float16 like the second layer -- ONE k-step of 32 and three
  // matrix instructions of 16 cycles per 16 x 16 tile instead of L / 4 = 7 exact-f32 ones of 32 cycles (round 6, late: a build without
  // the phase showed the first layers costing 5.8 of C5's 15.2 us per tick; with one k-step of seven 11.9).  The other precisions and the
  // smaller observations keep the exact-f32 form (at N = 4 the split of the inputs costs what the three k-steps do).
  static constexpr bool kF16Oth = 4 * L4 >= CM3_F16OTH_MIN && 4 * L4 <= 32;   // (N = 5 .. 9; N = 10's 36 inputs would need a second k-step)
  float bs[2], bias_s[4];       // bias of units 16w + 4 (l>>4) + reg: the TRANSPOSED C tile holds four units of one row per lane
  float bo[2][L4], bias_o[2][4];
  f16x8 boh[2], bol[2];         // kF16Oth: A[i = unit l&15][k = 8 (l>>4) + q] = W_others[k][unit], hi and scaled lo parts (kLoScale), 0 for k >= L
};

template <int N, typename T>
__device__ __forceinline__ void actor_first_b(const T *self_tab, const T *oth_tab, int w, int lane, bool stage2,
                                              ActorFirstB<N> &f) {
  using PL = PackLayout<N>;
  const int col = lane & 15, hi = lane >> 4;
  const int unit = 16 * w + col;
#pragma unroll
  for (int s = 0; s < 2; ++s) f.bs[s] = (4 * s + hi < 6) ? self_tab[unit * PL::SW + 4 * s + hi] : 0.0f;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) f.bias_s[reg] = self_tab[(16 * w + 4 * hi + reg) * PL::SW + 6];
#pragma unroll
  for (int cq = 0; cq < 2; ++cq) {
    const int uo = 32 * w + 16 * cq + col;
#pragma unroll
    for (int s = 0; s < ActorFirstB<N>::L4; ++s) f.bo[cq][s] = stage2 ? oth_tab[uo * PL::OW + 4 * s + hi] : 0.0f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) f.bias_o[cq][reg] = stage2 ? oth_tab[(32 * w + 16 * cq + 4 * hi + reg) * PL::OW + PL::L] : 0.0f;
    if constexpr (ActorFirstB<N>::kF16Oth) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = 8 * hi + q;
        const float wv = (stage2 && k < PL::L) ? (float)oth_tab[uo * PL::OW + (k < PL::L ? k : 0)] : 0.0f;
        const _Float16 h = (_Float16)wv;
        f.boh[cq][q] = h;
        f.bol[cq][q] = (_Float16)((wv - (float)h) * kLoScale);
      }
    }
  }
}

// xs tile + tables ready and synchronised on entry; h2s ready and synchronised on exit (3 barriers inside, 2 where h2 has
// storage of its own: ActorGeom::H2SEP).  The lane's
// phase-A B operands are read from the LDS tables ONCE, up front (measured -0.9 % per launch against reading them inside each
// block: 8.80 -> 8.72 us at 16 384 rows, same box).
template <int N, int PREC, int RT>
__device__ __forceinline__ void actor_mlp(const ActorLds<N, PREC, RT> &lds, const ActorB<N, PREC> &b, const ActorFirstB<N> &f1, int w,
                                          int lane, bool stage2) {
  using G = ActorGeom<N, PREC, RT>;
  constexpr bool BF16 = G::BF16;
  constexpr int L = G::L, KU = G::KU;
  const int col = lane & 15, hi = lane >> 4, c0 = 16 * w;
  (void)stage2;   // (f1 = the lane's first-layer operands, actor_first_b: read from the LDS tables ONCE per launch by the caller)
  // ---- phase A: dense(6 -> 64) units [16w, 16w+16) and dense(L -> 128) units [32w, 32w+32) (networks.py:520-529) ------
#ifndef CM3_PROBE_P_NO_PHASEA   // (probe builds only, tools/r6/policy_whatif.sh: what a part of the tick costs; results are wrong by construction)
  {
    float ax[RT][2], ao[RT][L / 4];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
      for (int s = 0; s < 2; ++s) ax[t][s] = (4 * s + hi < 6) ? lds.xs[16 * t + col][4 * s + hi] : 0.0f;
#pragma unroll
      for (int s = 0; s < L / 4; ++s) ao[t][s] = lds.xs[16 * t + col][6 + 4 * s + hi];
    }
    // TRANSPOSED tiles (round 3): C[i = unit][j = row] = sum_k W1[k][unit] x[row][k] -- the weight operand goes in as A, the input
    // tile as B (the per-lane registers are the same ones as for C[row][unit]; same products in the same k order, so the values are
    // bit-identical).  A lane then holds FOUR CONSECUTIVE UNITS of ONE row (row 16t + (l&15), units u0 + 4 (l>>4) + reg): one
    // ds_write_b64 per four float16 values (two for hi + lo), two per four float32 values -- 24 LDS stores per wave instead of 96 / 48.
    auto put4 = [&](int row, int unit0, const float (&h)[4]) {
      if constexpr (PREC == kPrecF16x3) {
        // hi = f16(h) (packed convert), lo' = f16((h - hi) * 2^11) as fma(hi, -2^11, h * 2^11) on the pair: exact before its single
        // rounding to float16 -- the same bits as subtract, scale, convert
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        uint2 vh, vl;
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          // Plain vector C, NOT inline assembly (round 4 had v_fma_mixlo / mixhi_f16 in asm statements here, one instruction less
          // per value): with the accumulators in architectural VGPRs an asm statement's registers can be ones a matrix instruction in
          // flight still reads or writes, and the hazard recogniser does not look into asm (how the inline relu failed, see
          // relu_f32).  No failure was traced to those two; this form cannot have one.  (hi as float) * -2^11 + h * 2^11 is exact in
          // float32 (the residual of an 11-bit rounding has at most 13 significant bits), so the one rounding is the conversion.
          const f2 hv = f2{h[2 * r2], h[2 * r2 + 1]};
          const h2 hp = __builtin_convertvector(hv, h2);
          const f2 res = __builtin_elementwise_fma(__builtin_convertvector(hp, f2), f2{-kLoScale, -kLoScale}, hv * kLoScale);
          const h2 lp = __builtin_convertvector(res, h2);
          uint32_t hw, lw;
          __builtin_memcpy(&hw, &hp, 4);
          __builtin_memcpy(&lw, &lp, 4);
          if (r2 == 0) { vh.x = hw; vl.x = lw; } else { vh.y = hw; vl.y = lw; }
        }
        *reinterpret_cast<uint2 *>(&lds.h1h[row][unit0]) = vh;
        *reinterpret_cast<uint2 *>(&lds.h1l[row][unit0]) = vl;
      } else if constexpr (BF16) {
        typedef __bf16 b4 __attribute__((ext_vector_type(4)));
        b4 vb;
#pragma unroll
        for (int r = 0; r < 4; ++r) vb[r] = (__bf16)h[r];
        *reinterpret_cast<b4 *>(&lds.h1b[row][unit0]) = vb;
      } else {
        *reinterpret_cast<float2 *>(&lds.h1s[row][unit0]) = make_float2(h[0], h[1]);
        *reinterpret_cast<float2 *>(&lds.h1s[row][unit0 + 2]) = make_float2(h[2], h[3]);
      }
    };
    // Per 16-unit column tile: all its matrix instructions first, every row tile into its OWN accumulator, s-major (two
    // instructions on the same accumulator are RT issues apart), then the epilogues (round 4: one accumulator shared by all
    // twelve tiles made them a serial chain of issue / wait / read-out).  The bias rides in as the accumulator's start value.
    {  // branch_self: units 16w .. 16w + 15
      f32x4 c[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) c[t] = f32x4{f1.bias_s[0], f1.bias_s[1], f1.bias_s[2], f1.bias_s[3]};
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < RT; ++t) c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.bs[s], ax[t][s], c[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        float h[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) h[reg] = relu_f32(c[t][reg]);
        put4(16 * t + col, 16 * w + 4 * hi, h);
      }
    }
    if constexpr (PREC == kPrecF16x3 && ActorFirstB<N>::kF16Oth) {
      // actor_others in split float16 (ActorFirstB::kF16Oth): B[k = 8 (l>>4) + q][j = l&15] = input k of row 16t + (l&15), split per tick
      // into hi = f16(x) and lo' = f16((x - hi) 2^11) like the second layer's activations (put4); inputs k >= L are masked to zero (the
      // reads run past the row), the weights there are zero too.  Two accumulators per tile: hi x hi on the bias, the small terms apart.
      typedef _Float16 h2 __attribute__((ext_vector_type(2)));
      typedef float f2 __attribute__((ext_vector_type(2)));
      f16x8 xh[RT], xl[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = lds.xs[16 * t + col][6 + 8 * hi + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = (8 * hi + q < L) ? x[q] : 0.0f;
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          const f2 xv = f2{x[2 * p2], x[2 * p2 + 1]};
          const h2 hp = __builtin_convertvector(xv, h2);
          const f2 res = __builtin_elementwise_fma(__builtin_convertvector(hp, f2), f2{-kLoScale, -kLoScale}, xv * kLoScale);
          const h2 lp = __builtin_convertvector(res, h2);
          xh[t][2 * p2] = hp[0]; xh[t][2 * p2 + 1] = hp[1];
          xl[t][2 * p2] = lp[0]; xl[t][2 * p2 + 1] = lp[1];
        }
      }
#pragma unroll
      for (int cq = 0; cq < 2; ++cq) {
        f32x4 d[RT], ds[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          d[t] = f32x4{f1.bias_o[cq][0], f1.bias_o[cq][1], f1.bias_o[cq][2], f1.bias_o[cq][3]};
          ds[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) ds[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1.bol[cq], xh[t], ds[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1.boh[cq], xh[t], d[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) ds[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f1.boh[cq], xl[t], ds[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          float h[4];
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) h[reg] = relu_f32(fmaf(ds[t][reg], kLoUnscale, d[t][reg]));
          put4(16 * t + col, kH1S + 32 * w + 16 * cq + 4 * hi, h);
        }
      }
    } else {
#pragma unroll
    for (int cq = 0; cq < 2; ++cq) {  // actor_others: two 16-unit tiles, units 32w + 16cq .. + 15 (stage 1: exactly 0, no others branch)
      f32x4 d[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) d[t] = f32x4{f1.bias_o[cq][0], f1.bias_o[cq][1], f1.bias_o[cq][2], f1.bias_o[cq][3]};
#ifdef CM3_PROBE_P_PHASEA_KS1   // (probe builds only: the exact-f32 others layer with ONE k-step)
      constexpr int KSO = 1;
#else
      constexpr int KSO = L / 4;
#endif
#pragma unroll
      for (int s = 0; s < KSO; ++s)
#pragma unroll
        for (int t = 0; t < RT; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f1.bo[cq][s], ao[t][s], d[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        float h[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) h[reg] = relu_f32(d[t][reg]);
        put4(16 * t + col, kH1S + 32 * w + 16 * cq + 4 * hi, h);
      }
    }
    }
  }
#endif
  CM3_STAMP(3, true);
  __syncthreads();
  CM3_STAMP(4, false);
  // ---- phase B: second layer on the matrix cores (networks.py:522-531: both matmuls, add_n) ---------------------------
  // NOT transposed like the first layer.  Tried (round 4, late: weights as A, one 16-byte h2 store per tile): no faster -- and it is
  // the form under which round 4 saw a handful of wrong rows per launch with two N = 8 workgroups on a CU.  Round 5 traced those to
  // the PHYSICS of the wave that shares the SIMD, not to this layer: a packed float32 multiply with a cross-half operand select
  // (v_pk_mul_f32 ... op_sel:[0,1], made by the SLP vectoriser out of the contact chain) returns a wrong low half in lanes 48..63
  // while the other wave executes v_mfma_f32_16x16x32_f16 -- ~100 x more often with the activations as the B operand than with this
  // order, never without these matrix instructions (profiles/r05_policy_fault.txt).  The build no longer contains that instruction
  // form (csrc/build.sh: -fno-slp-vectorize, tools/isa_lint.py); the operand order stays as it was measured.
  // tests/test_gpu_actor.py::test_policy_rollout_row_tile_rule_at_the_baseline_sizes is the test that caught it.
  f32x4 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#ifndef CM3_PROBE_P_NO_PHASEB
  if constexpr (PREC == kPrecF16x3) {
    f32x4 accs[RT];   // hi x lo' + lo' x hi, scaled by 2^11 (kLoScale)
#pragma unroll
    for (int t = 0; t < RT; ++t) accs[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < KU / 32; ++s) {
      // each product over all row tiles before the next, the small terms in their own accumulators: two MFMAs on the same
      // accumulator are at least 2 RT issues apart (back to back they wait for each other: measured on the Checkers actor)
      f16x8 ah[RT], al[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        ah[t] = *reinterpret_cast<const f16x8 *>(&lds.h1h[16 * t + col][32 * s + 8 * hi]);
        al[t] = *reinterpret_cast<const f16x8 *>(&lds.h1l[16 * t + col][32 * s + 8 * hi]);
      }
#pragma unroll
      for (int t = 0; t < RT; ++t) accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], b.bwh[s], accs[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], b.bwh[s], acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < RT; ++t) accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], b.bwl[s], accs[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) acc[t][reg] = fmaf(accs[t][reg], kLoUnscale, acc[t][reg]);   // (the scale is a power of two: exact)
  } else if constexpr (BF16) {
#pragma unroll
    for (int s = 0; s < KU / 32; ++s) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&lds.h1b[16 * t + col][32 * s + 8 * hi]);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.bwb[s], acc[t], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < KU / 4; ++s) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(lds.h1s[16 * t + col][4 * s + hi], b.bw[s], acc[t], 0, 0, 0);
    }
  }
#endif
  CM3_STAMP(5, true);
  if constexpr (!G::H2SEP) __syncthreads();  // all waves have consumed h1: its storage becomes h2
  // ---- h2 = relu(add_n + b) (networks.py:533-534): C tile -> LDS rows --------------------------------------------------
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) lds.h2s[16 * t + 4 * hi + reg][c0 + col] = relu_f32(acc[t][reg] + b.bias_h2);
  __syncthreads();
  CM3_STAMP(6, false);
}

// Final layer: wave w finishes rows [16w, 16w+16) on the matrix cores (round 3) -- one 16 x 16 tile, K = 64, TRANSPOSED like the
// first layer's tiles (round 4): C[i = action][j = row] = sum_k w_out[k][i] h2[row j][k]
//   A[i = l&15][k = l>>4] = w_out[4s + (l>>4)][i] for i < 5, else 0   (16 VGPRs per lane, ActorHeadB, loaded once per launch)
//   B[k = l>>4][j = l&15] = h2[16w + (l&15)][4s + (l>>4)]             one ds_read_b32 per k-step, all sixteen requested up front
// 16 v_mfma_f32_16x16x4_f32 = a k-ordered fmaf chain per logit (exact float32, like the other layers), bias as the start value.
// Lane l then holds actions 4 (l>>4) + reg of row 16w + (l&15): logits 0..3 of a row sit in its lane of the first 16, logit 4 in
// register 0 of the lane 16 further on -- ONE v_permlane16_swap brings it over, and lanes 0..15 continue with all five: softmax
// (networks.py:536-537), epsilon mix (alg_credit.py:119).  (Before: C[row][action] through a 2 KB LDS tile -- four stores, a
// wave-level hand-over and two loads on the path of every tick.  The first version gave every lane a quarter of the 64 units: 80
// broadcast LDS reads of w_out + 16 of h2 + 80 FMAs + 10 shuffles per lane -- 3.5k of the wave's ~14k cycles per tick;
// profiles/r03_actor_*.)  Only lanes 0..15 return probabilities; the other lanes' values are unspecified (finite).
struct ActorHeadB {
  float wo[kH2 / 4];
  float bias4[4];
};
__device__ __forceinline__ void actor_head_load(const float *wout, int lane, ActorHeadB &hb) {
  const int j = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int s = 0; s < kH2 / 4; ++s) hb.wo[s] = j < kA ? wout[(4 * s + hi) * kA + j] : 0.0f;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) hb.bias4[reg] = 4 * hi + reg < kA ? wout[kH2 * kA + 4 * hi + reg] : 0.0f;
}

__device__ __forceinline__ void actor_head_probs(const float (*h2s)[kH2W], const ActorHeadB &hb, int w, int lane, float eps,
                                                 float (&pr)[kA]) {
  const int col = lane & 15, hi = lane >> 4;
  // two accumulators (even / odd k-steps), added at the end: sixteen instructions on ONE accumulator each waited for its
  // predecessor's eight passes (round 4: ~200 of the head's ~2000 cycles); the bias starts the even chain
  f32x4 acc = f32x4{hb.bias4[0], hb.bias4[1], hb.bias4[2], hb.bias4[3]}, acc1 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // all sixteen B operands first (the compiler otherwise reads two, waits, issues two instructions, eight times over: eight LDS
  // round trips in a row on the path every tick of the rollout waits for)
  float hx[kH2 / 4];
#pragma unroll
  for (int s = 0; s < kH2 / 4; ++s) hx[s] = h2s[16 * w + col][4 * s + hi];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < kH2 / 4; s += 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hb.wo[s], hx[s], acc, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hb.wo[s + 1], hx[s + 1], acc1, 0, 0, 0);
  }
  float o[kA];
#pragma unroll
  for (int a = 0; a < 4; ++a) o[a] = acc[a] + acc1[a];
  {
    // rows of 16 lanes: the odd rows of the first operand change places with the even rows of the second, so the second result's
    // lanes 0..15 hold what lanes 16..31 had -- action 4 of the same matrix row
    uint32_t x;
    __builtin_memcpy(&x, &o[0], 4);
    const auto sw = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const uint32_t y = sw[1];
    __builtin_memcpy(&o[4], &y, 4);
  }
  float m = o[0];
#pragma unroll
  for (int a = 1; a < kA; ++a) m = fmaxf(m, o[a]);
  float sum = 0.0f;
#pragma unroll
  for (int a = 0; a < kA; ++a) {
    // exp on the hardware unit (v_exp_f32, base 2, ~1 ulp): the argument is <= 0, the result in (0, 1], and a relative 1e-7 on
    // a probability is two orders of magnitude inside the 2e-5 the head is held to (libm's expf was ~12 instructions x 5 on the
    // path of the 16 head lanes of every wave)
    o[a] = __builtin_amdgcn_exp2f((o[a] - m) * 1.44269504088896340736f);
    sum += o[a];
  }
  // (hardware reciprocal, 1 ulp: the sum is in [1, 5]; the IEEE division was ten instructions on the same path)
  const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
  for (int a = 0; a < kA; ++a) pr[a] = (1.0f - eps) * (o[a] * inv) + eps / (float)kA;
}

// The value lanes 0..15 hold, in all four 16-lane rows (lane l gets lane l & 15's): two swaps, no LDS
__device__ __forceinline__ int bcast_row0(int v) {
  uint32_t x = (uint32_t)v;
  x = __builtin_amdgcn_permlane16_swap(x, x, false, false)[0];   // row 1 <- row 0 (and row 3 <- row 2)
  x = __builtin_amdgcn_permlane32_swap(x, x, false, false)[0];   // rows 2, 3 <- rows 0, 1
  return (int)x;
}

template <int N, int PREC> __global__ void CM3_MATRIX_KERNEL k_actor_particle(const ActorParams p) {
  using G = ActorGeom<N, PREC>;
  constexpr int L = G::L;
  CM3_ACTOR_LDS(N, PREC, lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t rows = (size_t)p.E * N;
  const size_t row_base = (size_t)blockIdx.x * 64;
  CM3_STAMP(0, false);

  // RNG key of the row this lane finishes in the head (row 16w + (l&15)): fetched now, used ~20k cycles later
  size_t hr = row_base + 16 * w + (lane & 15);
  const bool head_ok = hr < rows && (lane >> 4) == 0;
  hr = hr < rows ? hr : rows - 1;
  const size_t he = hr / N;
  const int hi_agent = (int)(hr - he * N);
  const int head_steps = p.meta[2 * he];
  const uint32_t head_episode = (uint32_t)p.episode[he];

  // Kernel entry: every global request first (tables, wave 0's input rows, the W2 slice), then the LDS stores.
  // Measured and REJECTED in round 2 (tools/probes/actor_timeline.hip, 16 384 rows, same box, three repeats):
  // first-layer operands straight from global memory into registers instead of the table copy (+9 %: the entry is bound by the
  // request throughput of 256 workgroups reading the same few KB of L2, and narrow per-lane requests are worse than the wide
  // copy), the Philox draw before the first barrier (+0.5 %), input rows staged by all four waves (+-0), both (+1.1 %).
  ActorTabRegs<N> tab;
  actor_tables_fetch<N>(p.packed, tid, tab);
  CM3_STAMP(8, false);
  // the 64 input rows [v_obs(4) | v_goal(2) | obs_others(L)], one row per lane: wave 0 stages them into LDS; the other waves
  // request the same (coalesced, cached) lines rather than branch around the loads
  float4 in_s, in_o[L / 4];
  float2 in_g;
  {
    const size_t r = row_base + lane;
    const size_t rc = r < rows ? r : rows - 1;
    const size_t e = rc / N;
    const int i = (int)(rc - e * N);
    in_s = reinterpret_cast<const float4 *>(p.state)[(size_t)i * p.E + e];
    in_g = reinterpret_cast<const float2 *>(p.goals)[(size_t)i * p.E + e];
    const float4 *o4 = reinterpret_cast<const float4 *>(p.obs_others + rc * L);
#pragma unroll
    for (int k = 0; k < L / 4; ++k) in_o[k] = o4[k];
  }
  CM3_STAMP(9, false);
  ActorB<N, PREC> b;
  actor_load_b<N, PREC>(p.packed, w, lane, b);
  CM3_STAMP(10, false);
  // ... everything is on its way: now the LDS side
  actor_tables_store<N, PREC, 4>(lds, tid, tab);
  if (w == 0) {
    lds.xs[lane][0] = in_s.x; lds.xs[lane][1] = in_s.y; lds.xs[lane][2] = in_s.z; lds.xs[lane][3] = in_s.w;
    lds.xs[lane][4] = in_g.x; lds.xs[lane][5] = in_g.y;
#pragma unroll
    for (int k = 0; k < L / 4; ++k) {
      lds.xs[lane][6 + 4 * k + 0] = in_o[k].x; lds.xs[lane][6 + 4 * k + 1] = in_o[k].y;
      lds.xs[lane][6 + 4 * k + 2] = in_o[k].z; lds.xs[lane][6 + 4 * k + 3] = in_o[k].w;
    }
  }
  CM3_STAMP(1, true);
  __syncthreads();
  CM3_STAMP(2, false);
  ActorHeadB hb;
  actor_head_load(lds.wout, lane, hb);      // (the tables are in LDS: synchronised above)
  ActorFirstB<N> f1;
  actor_first_b<N, float>(&lds.ws_self[0][0], &lds.ws_oth[0][0], w, lane, p.stage > 1, f1);
  actor_mlp<N, PREC, 4>(lds, b, f1, w, lane, p.stage > 1);
  float pr[kA];
  // the uniform first: its Philox rounds are VALU work that can issue between the head's dependent MFMAs
  const float u = actor_uniform(p.seed, (uint64_t)(p.env_id_base + (int64_t)he), head_episode, head_steps, hi_agent);
  actor_head_probs(lds.h2s, hb, w, lane, p.eps_dev ? *p.eps_dev : p.eps, pr);
  const int act = actor_pick(pr, u);
  if (head_ok) {
    p.actions[hr] = act;
    if (p.probs) {
#pragma unroll
      for (int a = 0; a < kA; ++a) p.probs[hr * kA + a] = pr[a];
    }
  }
  CM3_STAMP(7, true);
}

template <int N> static int actor_launch(const ActorParams &p, hipStream_t s) {
  const size_t rows = (size_t)p.E * N;
  const unsigned blocks = (unsigned)((rows + 63) / 64);
  if (p.bf16 == kPrecF16x3)
    hipLaunchKernelGGL((k_actor_particle<N, kPrecF16x3>), dim3(blocks), dim3(256), 0, s, p);
  else if (p.bf16 == kPrecBf16)
    hipLaunchKernelGGL((k_actor_particle<N, kPrecBf16>), dim3(blocks), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((k_actor_particle<N, kPrecF32>), dim3(blocks), dim3(256), 0, s, p);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

}  // namespace cm3

namespace cm3 {
static int actor_check_desc(const cm3_actor_particle_desc *d) {
  CM3_REQUIRE(d, "null desc");
  CM3_REQUIRE(d->n_agents >= 1 && d->n_agents <= CM3_MAX_AGENTS, "n_agents must be in 1..%d", CM3_MAX_AGENTS);
  CM3_REQUIRE(d->n_h1_self == kH1S && d->n_h1_others == kH1O && d->n_h2 == kH2 && d->n_actions == kA,
              "supported actor widths are 64/128/64/5 (config.json nn block); got %d/%d/%d/%d", d->n_h1_self,
              d->n_h1_others, d->n_h2, d->n_actions);
  return CM3_OK;
}
template <int N> static size_t packed_floats() { return (size_t)PackLayout<N>::kTotal; }
static size_t packed_floats_for(int n) {
  switch (n) {
    case 1: return packed_floats<1>();
    case 2: return packed_floats<2>();
    case 3: return packed_floats<3>();
    case 4: return packed_floats<4>();
    case 5: return packed_floats<5>();
    case 6: return packed_floats<6>();
    case 7: return packed_floats<7>();
    case 8: return packed_floats<8>();
    case 9: return packed_floats<9>();
    case 10: return packed_floats<10>();
  }
  return 0;
}
template <int N> static int pack_launch(const ActorParams &p, float *out, hipStream_t s) {
  hipLaunchKernelGGL((k_actor_pack<N>), dim3(32), dim3(256), 0, s, p, out);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}
}  // namespace cm3

#ifndef CM3_NO_ENTRY_POINTS
extern "C" size_t cm3_actor_particle_packed_bytes(int32_t n_agents) {
  return cm3::packed_floats_for(n_agents) * sizeof(float);
}

extern "C" int cm3_actor_particle_pack(const cm3_actor_particle_desc *d, const cm3_actor_particle_weights *wt,
                                       void *packed, void *stream) {
  using namespace cm3;
  int rc = actor_check_desc(d);
  if (rc != CM3_OK) return rc;
  CM3_REQUIRE(wt && packed, "null weights / packed buffer");
  CM3_REQUIRE(wt->w_self && wt->b_self && wt->w_self_h2 && wt->b_h2 && wt->w_out && wt->b_out, "missing weights");
  if (d->stage > 1) CM3_REQUIRE(wt->w_others && wt->b_others && wt->w_others_h2, "stage 2 needs the others branch");
  ActorParams p;
  memset(&p, 0, sizeof(p));
  p.stage = d->stage;
  p.w_self = wt->w_self; p.b_self = wt->b_self; p.w_self_h2 = wt->w_self_h2;
  p.w_oth = wt->w_others; p.b_oth = wt->b_others; p.w_oth_h2 = wt->w_others_h2;
  p.b_h2 = wt->b_h2; p.w_out = wt->w_out; p.b_out = wt->b_out;
  hipStream_t s = (hipStream_t)stream;
  switch (d->n_agents) {
    case 1: return pack_launch<1>(p, (float *)packed, s);
    case 2: return pack_launch<2>(p, (float *)packed, s);
    case 3: return pack_launch<3>(p, (float *)packed, s);
    case 4: return pack_launch<4>(p, (float *)packed, s);
    case 5: return pack_launch<5>(p, (float *)packed, s);
    case 6: return pack_launch<6>(p, (float *)packed, s);
    case 7: return pack_launch<7>(p, (float *)packed, s);
    case 8: return pack_launch<8>(p, (float *)packed, s);
    case 9: return pack_launch<9>(p, (float *)packed, s);
    case 10: return pack_launch<10>(p, (float *)packed, s);
  }
  return fail(CM3_ERR_INVALID, "n_agents %d unsupported", d->n_agents);
}

extern "C" int cm3_actor_particle_f32(const cm3_actor_particle_desc *d, const cm3_actor_particle_weights *wt,
                                      const cm3_actor_particle_bufs *b, void *stream) {
  using namespace cm3;
  CM3_REQUIRE(d && wt && b, "null desc/weights/bufs");
  CM3_REQUIRE(d->n_envs > 0, "n_envs must be positive");
  CM3_REQUIRE(d->n_agents >= 1 && d->n_agents <= CM3_MAX_AGENTS, "n_agents must be in 1..%d", CM3_MAX_AGENTS);
  CM3_REQUIRE(d->n_h1_self == kH1S && d->n_h1_others == kH1O && d->n_h2 == kH2 && d->n_actions == kA,
              "supported actor widths are 64/128/64/5 (config.json nn block); got %d/%d/%d/%d", d->n_h1_self,
              d->n_h1_others, d->n_h2, d->n_actions);
  CM3_REQUIRE(d->epsilon >= 0.0f && d->epsilon <= 1.0f, "epsilon must be in [0,1]");
  CM3_REQUIRE(d->precision >= 0 && d->precision <= 2, "precision must be 0 (float32), 1 (bf16 second layer) or 2 (split float16)");
  CM3_REQUIRE(wt->packed, "weights->packed is NULL: run cm3_actor_particle_pack once per weight update");
  CM3_REQUIRE(b->obs_others && b->state && b->goals && b->meta && b->episode && b->actions, "missing buffers");
  ActorParams p;
  memset(&p, 0, sizeof(p));
  p.E = d->n_envs;
  p.stage = d->stage;
  p.eps = d->epsilon;
  p.eps_dev = b->epsilon_dev;
  p.bf16 = d->precision;
  p.env_id_base = d->env_id_base;
  p.seed = d->seed;
  p.obs_others = (const float *)b->obs_others;
  p.state = (const float *)b->state;
  p.goals = (const float *)b->goals;
  p.meta = b->meta;
  p.episode = b->episode;
  p.actions = b->actions;
  p.probs = b->probs;
  p.packed = (const float *)wt->packed;
  hipStream_t s = (hipStream_t)stream;
  switch (d->n_agents) {
    case 1: return actor_launch<1>(p, s);
    case 2: return actor_launch<2>(p, s);
    case 3: return actor_launch<3>(p, s);
    case 4: return actor_launch<4>(p, s);
    case 5: return actor_launch<5>(p, s);
    case 6: return actor_launch<6>(p, s);
    case 7: return actor_launch<7>(p, s);
    case 8: return actor_launch<8>(p, s);
    case 9: return actor_launch<9>(p, s);
    case 10: return actor_launch<10>(p, s);
  }
  return fail(CM3_ERR_INVALID, "n_agents %d unsupported", d->n_agents);
}
#endif  // CM3_NO_ENTRY_POINTS

// On-device actor for Checkers: forward pass + epsilon-mixed categorical sampling for all E*N agent rows in one launch
// (Checkers variant).
//
// Reference being replaced:
//   networks.convnet_1 + networks.actor_checkers   /root/reference/alg/networks.py:67-75, :549-578
//   probs = (1-eps) probs + eps/5, multinomial     /root/reference/alg/alg_credit_checkers.py:107-113
//   Alg.run_actor (actions_prev -> one-hot)        /root/reference/alg/alg_credit_checkers.py:229-253
// Widths (config_checkers_stage{1,2}.json "nn"): conv 3x3x3 -> 6 filters over the 5x5x3 window, conv_linear 150 -> 32,
// concat(32 + 4 + 5 + 2 = 43) -> 256, [256 | 256] -> 256, 256 -> 5.  153 k MACs per agent row, 86 % of them in the
// 512 -> 256 layer: unlike the env kernels this IS contraction work, so every layer runs on the matrix cores with the
// exact-f32 MFMA (v_mfma_f32_16x16x4_f32; float32 products and accumulation like the TF graph).
//
// Mapping: a 256-thread workgroup (4 waves, one per SIMD of a CU) owns 64 agent rows (row = e*N + i); activations live in
// LDS as [64][K + 4] float tiles (row stride = 4 mod 64 words: the 16-byte A-operand reads of 16 rows hit 16 distinct
// bank quads).  Every layer is the same tile loop (gemm_tiles): for a k-group of 16 inputs, one ds_read_b128 per 16-row
// tile gives the A operands of four MFMA k-steps, one 16-byte global load per 16-column tile gives the matching B
// operands from the PACKED weights (cm3_actor_checkers_pack: per-lane MFMA order, so a wave's load is one contiguous
// KB; the packed weights are 0.7 MB and stay in L2).  k-step j of group g contracts inputs {16g + 4h + j : h = 0..3} --
// any partition of K into fours is a valid summation order.
//   conv      : the 3x3 SAME convolution as a [80 x 160] Toeplitz matrix (zeros for out-of-window taps and padding)
//   precision = 1 (opt-in, not the parity path): the first-layer activations are stored as bf16 and the two 256 x 256
//               matrices run on v_mfma_f32_16x16x32_bf16 (f32 accumulation, 16x the f32 MFMA rate); everything else stays f32.
//   precision = 2 (split float16, a parity path; k_ck_actor_x3): EVERY layer as hi * hi + hi * lo + lo * hi on
//               v_mfma_f32_16x16x32_f16 with activations and weights split into float16 hi + lo (22 significand bits per factor).
//               Limit: the residual lo = f16(x - hi) is NOT kept scaled here (the particle actor's is, actor.hip
//               kLoScale: it has the accumulator registers to spare, this kernel's 4 x 4 tiles do not), so for |x| < 0.125 it
//               falls into float16's subnormals: an ABSOLUTE error of ~3e-8 per factor instead of a relative 2^-22.  Held to the
//               2e-5 bound with second-layer weights of ~6e-5 against first-layer activations of ~1e2
//               (tests/test_gpu_actor_checkers.py::test_split_precision_small_weights_large_activations); beyond
//               |w| |x| ~ 1e3 / 3e-8 products the bound is not guaranteed, and float16 saturates at 65504: use precision = 0 there.
//   wave tiles: conv 2 row x 5 col tiles, conv_linear 2 x 1, branch_self / branch_others / h2: 4 x 4 (all 64 rows x 64
//               columns per wave, 64 accumulator VGPRs), actor_out: wave w finishes rows [16w, 16w+16).
#include "actor_common.h"

namespace cm3 {

namespace ck_actor {
constexpr int kConvF = 6, kLin = 32, kH1 = 256, kH2 = 256;
constexpr int kObs = 75;                       // 5 x 5 x 3 window
constexpr int kKConv = 80, kNConv = 160;       // 75 -> 80 inputs, 150 -> 160 outputs
constexpr int kKLin = 160, kNLin = 32;
constexpr int kKSelf = 48, kCat = 43;          // 32 + 4 + 5 + 2 -> 48
constexpr int kKOth = 16;                      // 2 (N-1) <= 14 -> 16
// LDS row strides (floats): K + 4
constexpr int kLdX0 = kKConv + 4, kLdC1 = kNConv + 4, kLdX2 = kKSelf + 4, kLdXO = kKOth + 4, kLdH = kH1 + 4;

// packed weights (floats).  B tiles: [col tile][k group][lane][4]; biases follow each matrix, padded to the tile width.
constexpr int tiles(int k, int n) { return (n / 16) * (k / 16) * 256; }
constexpr int kPConv = 0;
constexpr int kPConvB = kPConv + tiles(kKConv, kNConv);
constexpr int kPLin = kPConvB + kNConv;
constexpr int kPLinB = kPLin + tiles(kKLin, kNLin);
constexpr int kPSelf = kPLinB + kNLin;
constexpr int kPSelfB = kPSelf + tiles(kKSelf, kH1);
constexpr int kPOth = kPSelfB + kH1;
constexpr int kPOthB = kPOth + tiles(kKOth, kH1);
constexpr int kPH2S = kPOthB + kH1;
constexpr int kPH2O = kPH2S + tiles(kH1, kH2);
constexpr int kPH2B = kPH2O + tiles(kH1, kH2);
constexpr int kPOut = kPH2B + kH2;
constexpr int kPOutB = kPOut + tiles(kH2, 16);
constexpr int kPF32 = kPOutB + 16;
// bf16 copies of the two 256 x 256 matrices for precision = 1: [col tile 16][k-step 8][lane 64][8 bf16], two per float slot
constexpr int kPH2Sb = kPF32;
constexpr int kPH2Ob = kPH2Sb + kH1 * kH2 / 2;
// float16 hi / lo copies of the same two matrices for precision = 2 (w = hi + lo + O(2^-22 w)), same tile order
constexpr int kPH2Sh = kPH2Ob + kH1 * kH2 / 2;
constexpr int kPH2Sl = kPH2Sh + kH1 * kH2 / 2;
constexpr int kPH2Oh = kPH2Sl + kH1 * kH2 / 2;
constexpr int kPH2Ol = kPH2Oh + kH1 * kH2 / 2;
// ... and of the five small matrices (precision = 2 runs every layer in split float16): K padded to a multiple of 32
constexpr int kKConvX = 96, kKSelfX = 64, kKOthX = 32;
constexpr int kXConvH = kPH2Ol + kH1 * kH2 / 2;
constexpr int kXConvL = kXConvH + kKConvX * kNConv / 2;
constexpr int kXLinH = kXConvL + kKConvX * kNConv / 2;
constexpr int kXLinL = kXLinH + kKLin * kNLin / 2;
constexpr int kXSelfH = kXLinL + kKLin * kNLin / 2;
constexpr int kXSelfL = kXSelfH + kKSelfX * kH1 / 2;
constexpr int kXOthH = kXSelfL + kKSelfX * kH1 / 2;
constexpr int kXOthL = kXOthH + kKOthX * kH1 / 2;
constexpr int kXOutH = kXOthL + kKOthX * kH1 / 2;
constexpr int kXOutL = kXOutH + kH2 * 16 / 2;
constexpr int kPTotal = kXOutL + kH2 * 16 / 2;
// two agents, stage 2: h2's accumulators after the others branch for each of the 7 x 13 cells the other agent can stand on
// (k_ck_actor_others_table; float32 [96][256], rows 91..95 unused)
constexpr int kPOthTab = kPTotal;
constexpr int kPAll = kPOthTab + 96 * kH2;
// float16 LDS planes of precision = 2: row strides in halfwords, K + 8 (= 4 x odd words: conflict-free 16-byte A reads)
// Row strides of the float16 / bf16 planes: K + 16 halfwords = 8 dwords mod 64.  (Until round 6: K + 8 = 4 dwords mod 64, "four
// times an odd number of words", which is conflict-free for 16 CONSECUTIVE lanes -- but ds_read_b128 serves a wave in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS table): lanes (row l & 15, k chunk l >> 4) of one group
// then hit bank 4 (row + chunk), where row 12 / chunk 0 and row 11 / chunk 1 collide -- EVERY A-operand read took two LDS cycles
// per group, SQ_LDS_BANK_CONFLICT was 45 % of SQ_LDS_IDX_ACTIVE (profiles/r06_pmc_ck_policy_summary.txt).  With 8 dwords per
// row the eight rows of a group's chunk-0 lanes and of its chunk-1 lanes cover the even and the odd 16-byte units.)
#ifndef CM3_CK_LD_PAD
#define CM3_CK_LD_PAD 16   // (macro: same-box A/B builds, tools/r6/ck_ld_pad_ab.sh)
#endif
constexpr int kLhX0 = kKConvX + CM3_CK_LD_PAD, kLhC1 = kNConv + CM3_CK_LD_PAD, kLhX2 = kKSelfX + CM3_CK_LD_PAD, kLhXO = kKOthX + CM3_CK_LD_PAD;
constexpr int kLdHb = kH1 + CM3_CK_LD_PAD;  // bf16 / f16 activation row: 272 halfwords = 544 B
}  // namespace ck_actor

struct CkActorParams {
  int E, N, stage, Lo;
  float eps;
  int64_t env_id_base;
  uint64_t seed;
  int obst_stride;  // bytes between envs of obs_self_t
  const int8_t *obs_self_t;
  const double *obs_self_v, *obs_others;
  const uint8_t *goals;
  const int32_t *actions_prev, *steps, *episode;
  const uint8_t *prev_done;  // optional uint8 [E]: envs whose previous tick ended an episode feed actions_prev = 0
  const float *eps_dev;      // optional: epsilon read from the device at launch
  int32_t *actions;
  float *probs;
  const float *packed;
  // pack kernel only (TensorFlow shapes)
  const float *conv_w, *conv_b, *lin_w, *lin_b, *self_w, *self_b, *w_self_h2, *oth_w, *oth_b, *w_oth_h2, *b_h2, *out_w,
      *out_b;
};

// ---- pack: TF-shaped weights -> per-lane MFMA order ------------------------------------------------------------------
// element (k, n) of the logical [K][Ncols] matrix of each layer, zero outside the real extent
__device__ __forceinline__ float ck_conv_toeplitz(const CkActorParams &p, int k, int n) {
  using namespace ck_actor;
  if (k >= kObs || n >= 25 * kConvF) return 0.0f;
  const int pin = k / 3, ch = k - 3 * pin, r = pin / 5, c = pin - 5 * r;          // input (r, c, ch)
  const int pout = n / kConvF, f = n - kConvF * pout, ro = pout / 5, co = pout - 5 * ro;  // output (ro, co, f)
  const int dr = r - ro + 1, dc = c - co + 1;  // out[ro][co] += x[ro + dr - 1][co + dc - 1] w[dr][dc]
  if (dr < 0 || dr > 2 || dc < 0 || dc > 2) return 0.0f;
  return p.conv_w[((dr * 3 + dc) * 3 + ch) * kConvF + f];
}

#ifndef CM3_NO_ENTRY_POINTS   // (non-template kernels: one definition per library; policy_checkers.hip includes this file for the device functions)
__global__ void __launch_bounds__(256) k_ck_actor_pack(const CkActorParams p, float *out) {
  using namespace ck_actor;
  const bool stage2 = p.stage > 1;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < kPTotal; t += gridDim.x * 256) {
    if (t >= kXConvH) {  // float16 hi / lo of the small matrices: [col tile][k-step][lane][8], k = 32 st + 8 (l >> 4) + q
      int base, KS, layer;
      if (t < kXLinH) { base = kXConvH; KS = kKConvX / 32; layer = 0; }
      else if (t < kXSelfH) { base = kXLinH; KS = kKLin / 32; layer = 1; }
      else if (t < kXOthH) { base = kXSelfH; KS = kKSelfX / 32; layer = 2; }
      else if (t < kXOutH) { base = kXOthH; KS = kKOthX / 32; layer = 3; }
      else { base = kXOutH; KS = kH2 / 32; layer = 6; }
      const int ncols = layer == 0 ? kNConv : (layer == 1 ? kNLin : (layer == 6 ? 16 : kH1));
      const int half = KS * 32 * ncols / 2;            // floats per plane
      const bool low = (t - base) >= half;
      _Float16 pair[2];
      for (int h = 0; h < 2; ++h) {
        const int eidx = 2 * (t - base - (low ? half : 0)) + h;
        const int q = eidx & 7, lane = (eidx >> 3) & 63, rest = eidx >> 9, st = rest % KS, ct = rest / KS;
        const int k = 32 * st + 8 * (lane >> 4) + q, n = 16 * ct + (lane & 15);
        float wv;
        switch (layer) {
          case 0: wv = ck_conv_toeplitz(p, k, n); break;
          case 1: wv = k < 25 * kConvF ? p.lin_w[k * kLin + n] : 0.0f; break;
          case 2: wv = k < kCat ? p.self_w[k * kH1 + n] : 0.0f; break;
          case 3: wv = (stage2 && k < p.Lo) ? p.oth_w[k * kH1 + n] : 0.0f; break;
          default: {   // actor_out: k-step st = the wave that owns units [32 st, 32 st + 32); slot 8 hi + q of it is unit
                       // 32 st + 4 hi + q (q < 4: the lane's first column tile) or 32 st + 16 + 4 hi + q - 4 (its second)
            const int hi4 = lane >> 4, unit = 32 * st + (q < 4 ? 4 * hi4 + q : 16 + 4 * hi4 + q - 4);
            wv = n < kA ? p.out_w[unit * kA + n] : 0.0f;
            break;
          }
        }
        const _Float16 hi = (_Float16)wv;
        pair[h] = low ? (_Float16)(wv - (float)hi) : hi;
      }
      float v;
      __builtin_memcpy(&v, pair, 4);
      out[t] = v;
      continue;
    }
    if (t >= kPH2Sh) {  // float16 hi / lo B operands of v_mfma_f32_16x16x32_f16, the layout of the bf16 copies below
      const int sec = (t - kPH2Sh) / (kH1 * kH2 / 2);   // 0: self hi, 1: self lo, 2: others hi, 3: others lo
      const bool oth = sec >= 2, low = (sec & 1) != 0;
      const float *src = oth ? p.w_oth_h2 : p.w_self_h2;
      _Float16 pair[2];
      for (int h = 0; h < 2; ++h) {
        const int eidx = 2 * (t - kPH2Sh - sec * (kH1 * kH2 / 2)) + h;
        const int q = eidx & 7, lane = (eidx >> 3) & 63, st = (eidx >> 9) & 7, ct = eidx >> 12;
        const int k = 32 * st + 8 * (lane >> 4) + q, n = 16 * ct + (lane & 15);
        const float wv = (oth && !stage2) ? 0.0f : src[k * kH2 + n];
        const _Float16 hi = (_Float16)wv;
        pair[h] = low ? (_Float16)(wv - (float)hi) : hi;
      }
      float v;
      __builtin_memcpy(&v, pair, 4);
      out[t] = v;
      continue;
    }
    if (t >= kPF32) {  // bf16 B operands of v_mfma_f32_16x16x32_bf16: B[k = 32 s + 8 (l >> 4) + q][n = 16 ct + (l & 15)]
      const bool oth = t >= kPH2Ob;
      const float *src = oth ? p.w_oth_h2 : p.w_self_h2;
      __bf16 pair[2];
      for (int h = 0; h < 2; ++h) {
        const int eidx = 2 * (t - (oth ? kPH2Ob : kPH2Sb)) + h;
        const int q = eidx & 7, lane = (eidx >> 3) & 63, st = (eidx >> 9) & 7, ct = eidx >> 12;
        const int k = 32 * st + 8 * (lane >> 4) + q, n = 16 * ct + (lane & 15);
        pair[h] = (__bf16)((oth && !stage2) ? 0.0f : src[k * kH2 + n]);
      }
      float v;
      __builtin_memcpy(&v, pair, 4);
      out[t] = v;
      continue;
    }
    // which section?
    int base, K, layer;
    if (t < kPConvB) { base = kPConv; K = kKConv; layer = 0; }
    else if (t < kPLin) { base = kPConvB; K = 0; layer = 10; }
    else if (t < kPLinB) { base = kPLin; K = kKLin; layer = 1; }
    else if (t < kPSelf) { base = kPLinB; K = 0; layer = 11; }
    else if (t < kPSelfB) { base = kPSelf; K = kKSelf; layer = 2; }
    else if (t < kPOth) { base = kPSelfB; K = 0; layer = 12; }
    else if (t < kPOthB) { base = kPOth; K = kKOth; layer = 3; }
    else if (t < kPH2S) { base = kPOthB; K = 0; layer = 13; }
    else if (t < kPH2O) { base = kPH2S; K = kH1; layer = 4; }
    else if (t < kPH2B) { base = kPH2O; K = kH1; layer = 5; }
    else if (t < kPOut) { base = kPH2B; K = 0; layer = 14; }
    else if (t < kPOutB) { base = kPOut; K = kH2; layer = 6; }
    else { base = kPOutB; K = 0; layer = 15; }
    const int u = t - base;
    float v = 0.0f;
    if (layer < 10) {
      const int j = u & 3, lane = (u >> 2) & 63, gi = u >> 8, KG = K / 16, g = gi % KG, ct = gi / KG;
      const int k = 16 * g + 4 * (lane >> 4) + j, n = 16 * ct + (lane & 15);
      switch (layer) {
        case 0: v = ck_conv_toeplitz(p, k, n); break;
        case 1: v = k < 25 * kConvF ? p.lin_w[k * kLin + n] : 0.0f; break;
        case 2: v = k < kCat ? p.self_w[k * kH1 + n] : 0.0f; break;
        case 3: {  // position k holds input 4 (k & 3) + (k >> 2): k-step j of the MFMA group contracts inputs 4j .. 4j+3
          const int kin = 4 * (k & 3) + (k >> 2);
          v = (stage2 && kin < p.Lo) ? p.oth_w[kin * kH1 + n] : 0.0f;
          break;
        }
        case 4: v = p.w_self_h2[k * kH2 + n]; break;
        case 5: v = stage2 ? p.w_oth_h2[k * kH2 + n] : 0.0f; break;
        default: v = n < kA ? p.out_w[k * kA + n] : 0.0f; break;
      }
    } else {
      switch (layer) {
        case 10: v = u < 25 * kConvF ? p.conv_b[u % kConvF] : 0.0f; break;
        case 11: v = p.lin_b[u]; break;
        case 12: v = p.self_b[u]; break;
        case 13: v = stage2 ? p.oth_b[u] : 0.0f; break;
        case 14: v = p.b_h2[u]; break;
        default: v = u < kA ? p.out_b[u] : 0.0f; break;
      }
    }
    out[t] = v;
  }
}

#endif  // CM3_NO_ENTRY_POINTS

// ---- the tile loop -----------------------------------------------------------------------------------------------------
// acc[t][c] += A[16 (rt0 + t) .. +16][0 .. 16 KG) x B[.., 16 (ct0 + c) .. +16].  A: LDS, row stride lda floats;
// Bp: packed tiles of this layer.  B operands of group g + 1 are requested before the MFMAs of group g issue; those of
// group 0 (b0) are requested by the caller with load_b0 BEFORE the previous layer's epilogue and barrier, so that their
// L2 latency hides behind the LDS stores (1.5-3 k cycles per layer when it was exposed).
template <int CT, int KG>
__device__ __forceinline__ void load_b0(const float *Bp, int ct0, int lane, float4 (&b0)[CT]) {
#pragma unroll
  for (int c = 0; c < CT; ++c) b0[c] = (reinterpret_cast<const float4 *>(Bp) + ((size_t)(ct0 + c) * KG) * 64 + lane)[0];
}

template <int RT, int CT, int KG>
__device__ __forceinline__ void gemm_tiles(const float *A, int lda, int rt0, const float *Bp, int ct0, int lane,
                                           const float4 (&b0)[CT], f32x4 (&acc)[RT][CT]) {
  const int col = lane & 15, hi = lane >> 4;
  const float4 *bsrc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) bsrc[c] = reinterpret_cast<const float4 *>(Bp) + ((size_t)(ct0 + c) * KG) * 64 + lane;
  const float *arow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) arow[t] = A + (16 * (rt0 + t) + col) * lda + 4 * hi;
  float4 bcur[CT], bnext[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) bcur[c] = b0[c];
#pragma unroll
  for (int g = 0; g < KG; ++g) {
    if (g + 1 < KG) {
#pragma unroll
      for (int c = 0; c < CT; ++c) bnext[c] = bsrc[c][(g + 1) * 64];
    }
    float4 a[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const float4 *>(arow[t] + 16 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const float av = j == 0 ? a[t].x : (j == 1 ? a[t].y : (j == 2 ? a[t].z : a[t].w));
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const float bv = j == 0 ? bcur[c].x : (j == 1 ? bcur[c].y : (j == 2 ? bcur[c].z : bcur[c].w));
          acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t][c], 0, 0, 0);
        }
      }
    }
    if (g + 1 < KG) {
#pragma unroll
      for (int c = 0; c < CT; ++c) bcur[c] = bnext[c];
    }
  }
}

template <int RT, int CT> __device__ __forceinline__ void zero_tiles(f32x4 (&acc)[RT][CT]) {
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[t][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
}

// this lane's bias values (column l & 15 of each column tile); requested before the layer's MFMAs so that the epilogue
// does not wait for them
template <int CT> __device__ __forceinline__ void load_bias(const float *bias, int ct0, int lane, float (&b)[CT]) {
#pragma unroll
  for (int c = 0; c < CT; ++c) b[c] = bias[16 * (ct0 + c) + (lane & 15)];
}

// O[row][col] = relu(acc + bias[col]); C layout: col = l & 15, row = 4 (l >> 4) + reg
template <int RT, int CT>
__device__ __forceinline__ void store_relu(float *O, int ldo, int rt0, int ct0, const float (&bias)[CT], int lane,
                                           const f32x4 (&acc)[RT][CT]) {
  const int col = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const float b = bias[c];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        O[(16 * (rt0 + t) + 4 * hi + reg) * ldo + 16 * (ct0 + c) + col] = fmaxf(acc[t][c][reg] + b, 0.0f);
  }
}

// ---- precision = 1: the two 256 x 256 layers (86 % of the FLOPs) on v_mfma_f32_16x16x32_bf16, f32 accumulation ----------
// first-layer activations rounded to bf16 on their way to LDS
template <int RT, int CT>
__device__ __forceinline__ void store_relu_bf16(__bf16 *O, int ldo, int rt0, int ct0, const float (&bias)[CT], int lane,
                                                const f32x4 (&acc)[RT][CT]) {
  const int col = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        O[(16 * (rt0 + t) + 4 * hi + reg) * ldo + 16 * (ct0 + c) + col] = (__bf16)fmaxf(acc[t][c][reg] + bias[c], 0.0f);
  }
}

template <int CT> __device__ __forceinline__ void load_b0_bf16(const float *Bp, int ct0, int lane, uint4 (&b0)[CT]) {
#pragma unroll
  for (int c = 0; c < CT; ++c) b0[c] = (reinterpret_cast<const uint4 *>(Bp) + ((size_t)(ct0 + c) * 8) * 64 + lane)[0];
}

// acc[t][c] += A[64 rows][256] x B[256][16 (ct0 + c) .. +16]: A[i = l & 15][k = 32 s + 8 (l >> 4) + q] one ds_read_b128 per
// row tile and k-step, B one 16-byte load per column tile and k-step (next step requested before this step's MFMAs)
template <int CT>
__device__ __forceinline__ void gemm_tiles_bf16(const __bf16 *A, int lda, const float *Bp, int ct0, int lane,
                                                const uint4 (&b0)[CT], f32x4 (&acc)[4][CT]) {
  const int col = lane & 15, hi = lane >> 4;
  const uint4 *bsrc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) bsrc[c] = reinterpret_cast<const uint4 *>(Bp) + ((size_t)(ct0 + c) * 8) * 64 + lane;
  uint4 bcur[CT], bnext[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) bcur[c] = b0[c];
#pragma unroll
  for (int st = 0; st < 8; ++st) {
    if (st + 1 < 8) {
#pragma unroll
      for (int c = 0; c < CT; ++c) bnext[c] = bsrc[c][(st + 1) * 64];
    }
    bf16x8 a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const bf16x8 *>(A + (16 * t + col) * lda + 32 * st + 8 * hi);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        bf16x8 bv;
        __builtin_memcpy(&bv, &bcur[c], 16);
        acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], bv, acc[t][c], 0, 0, 0);
      }
    if (st + 1 < 8) {
#pragma unroll
      for (int c = 0; c < CT; ++c) bcur[c] = bnext[c];
    }
  }
}

// ---- precision = 2 (split float16): three float16 MFMAs per product -- (a_lo, b_hi) + (a_hi, b_lo) + (a_hi, b_hi), float32
// accumulation.  Activations and weights are split x = hi + lo with hi = (float16)x, lo = (float16)(x - hi): 22 of float32's 24
// significand bits per factor, the error class of the exact-f32 path (tests hold the same 2e-5) at 16/3 of its MFMA rate.
// softmax (networks.py:576), epsilon mix (alg_credit_checkers.py:112), sampling (:113): lane l < 16 takes row 16w + l
// pr = (1 - eps) softmax(logits) + eps / 5 (networks.py:576, alg_credit_checkers.py:112) -- one definition for the stand-alone actor and
// the whole-episode kernel (same bits)
__device__ __forceinline__ void ck_actor_probs(const float (&logits)[kA], float eps, float (&pr)[kA]) {
  float o[kA];
  float m = logits[0];
#pragma unroll
  for (int a = 1; a < kA; ++a) m = fmaxf(m, logits[a]);
  float sum = 0.0f;
#pragma unroll
  for (int a = 0; a < kA; ++a) {
    // exp and the reciprocal on the hardware units (v_exp_f32 base 2, v_rcp_f32: ~1 ulp each; arguments <= 0, sum in [1, 5]) like the
    // particle head (actor.hip): a relative 1e-7 on a probability held to 2e-5.  libm's expf + the IEEE division were ~110 of the 190
    // instructions of the head that every tick of the whole-episode kernel waits for.
    o[a] = __builtin_amdgcn_exp2f((logits[a] - m) * 1.44269504088896340736f);
    sum += o[a];
  }
  const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
  for (int a = 0; a < kA; ++a) pr[a] = (1.0f - eps) * (o[a] * inv) + eps / (float)kA;
}

__device__ __forceinline__ void ck_actor_head(const CkActorParams &p, const float (*sLG)[8], int w, int lane, size_t row_base,
                                              size_t rows) {
  const int N = p.N;
  if (lane < 16) {
    const size_t row = row_base + 16 * w + lane;
    if (row < rows) {
      float o[kA], pr[kA];
#pragma unroll
      for (int a = 0; a < kA; ++a) o[a] = sLG[16 * w + lane][a];
      ck_actor_probs(o, p.eps_dev ? *p.eps_dev : p.eps, pr);
      const size_t e = row / N;
      const int i = (int)(row - e * N);
      const int act = actor_sample(pr, p.seed, (uint64_t)(p.env_id_base + (int64_t)e), (uint32_t)p.episode[e], p.steps[e], i);
      p.actions[row] = act;
      if (p.probs) {
#pragma unroll
        for (int a = 0; a < kA; ++a) p.probs[row * kA + a] = pr[a];
      }
    }
  }
}

// BF16: the two 256 x 256 layers on the bf16 matrix cores (precision = 1, not a parity path); otherwise float32 throughout
template <bool BF16> __global__ void CM3_MATRIX_KERNEL k_ck_actor(const CkActorParams p) {
  using namespace ck_actor;
  // H: [64][260] first-layer activations / h2; before that it holds X0 [64][84] and C1 [64][164]
  __shared__ __attribute__((aligned(16))) float sH[64 * kLdH];
  __shared__ __attribute__((aligned(16))) float sX2[64 * kLdX2];
  __shared__ __attribute__((aligned(16))) float sXO[64 * kLdXO];
  __shared__ float sLG[64][8];
  float *sX0 = sH, *sC1 = sH + 64 * kLdX0;
  __bf16 *sHb = reinterpret_cast<__bf16 *>(sH);  // BF16: first-layer activations [64][264] bf16
  static_assert(64 * kLdX0 + 64 * kLdC1 <= 64 * kLdH, "X0 and C1 must fit into the H storage");

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N;
  const size_t rows = (size_t)p.E * N;
  const size_t row_base = (size_t)blockIdx.x * 64;
  const float *pk = p.packed;
  CM3_STAMP(0, false);

  float4 b_conv[5];
  load_b0<5, kKConv / 16>(pk + kPConv, 5 * (w >> 1), lane, b_conv);
  // ---- stage the inputs ------------------------------------------------------------------------------------------------
  // window bytes -> floats (t_obs_self; values in {-1, 0, 1})
  if ((p.obst_stride & 3) == 0 && (64 % N) == 0) {
    // env records are dword-aligned and the 64 rows are whole envs: read each record as dwords (38 per env at N = 2
    // instead of 150 byte loads), scatter the four bytes to their (row, k) slots.  (Byte loads: 14 k cycles.)
    const int epw = 64 / N, dpe = p.obst_stride >> 2, rec = N * kObs;
    const size_t e0 = row_base / N;
    for (int d = tid; d < epw * dpe; d += 256) {
      const int el = d / dpe, dd = d - el * dpe;
      size_t e = e0 + el;
      e = e < (size_t)p.E ? e : (size_t)p.E - 1;
      const uint32_t v = reinterpret_cast<const uint32_t *>(p.obs_self_t + e * (size_t)p.obst_stride)[dd];
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const int bb = 4 * dd + sb;
        if (bb < rec) {
          const int i = bb / kObs, k = bb - i * kObs;
          sX0[(el * N + i) * kLdX0 + k] = (float)(int8_t)(v >> (8 * sb));
        }
      }
    }
    for (int idx = tid; idx < 64 * (kKConv - kObs); idx += 256) {
      const int r = idx / (kKConv - kObs), k = kObs + idx - r * (kKConv - kObs);
      sX0[r * kLdX0 + k] = 0.0f;
    }
  } else {
    for (int idx = tid; idx < 64 * kKConv; idx += 256) {
      const int r = idx / kKConv, k = idx - r * kKConv;
      size_t row = row_base + r;
      row = row < rows ? row : rows - 1;
      const size_t e = row / N;
      const int i = (int)(row - e * N);
      sX0[r * kLdX0 + k] = k < kObs ? (float)p.obs_self_t[e * (size_t)p.obst_stride + (size_t)i * kObs + k] : 0.0f;
    }
  }
  if (tid < 64) {  // concat tail: v_obs_self (4), a_prev one-hot (5), v_goal one-hot (2); pad; v_obs_others
    size_t row = row_base + tid;
    row = row < rows ? row : rows - 1;
    const size_t e = row / N;
    const int i = (int)(row - e * N);
    float *x = &sX2[tid * kLdX2 + kLin];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = (float)p.obs_self_v[row * 4 + k];
    // a fresh episode starts from actions_prev = zeros (train_onpolicy.py:295)
    const int ap = (p.actions_prev && !(p.prev_done && p.prev_done[e])) ? p.actions_prev[row] : 0;
#pragma unroll
    for (int k = 0; k < kA; ++k) x[4 + k] = ap == k ? 1.0f : 0.0f;
    const int gl = p.goals[row];
    x[9] = gl == 0 ? 1.0f : 0.0f;
    x[10] = gl == 0 ? 0.0f : 1.0f;
#pragma unroll
    for (int k = kCat - kLin; k < kKSelf - kLin; ++k) x[k] = 0.0f;
    // v_obs_others: input k sits at position 4 (k & 3) + (k >> 2), so that MFMA k-step j contracts inputs 4j .. 4j+3 and
    // only ceil(Lo / 4) of the four steps of the group are issued
    float *xo = &sXO[tid * kLdXO];
    for (int q = 0; q < kKOth; ++q) {
      const int k = 4 * (q & 3) + (q >> 2);
      xo[q] = k < p.Lo ? (float)p.obs_others[row * p.Lo + k] : 0.0f;
    }
    (void)i;
  }
  CM3_STAMP(1, true);
  __syncthreads();
  CM3_STAMP(2, false);

  // ---- conv (Toeplitz) : X0 [64][80] -> C1 [64][160], relu -----------------------------------------------------------------
  float4 b_lin[1], b_self[4], b_h2[4], b_oth[4], b_out[1];
  uint4 b_h2b[4];
  {
    f32x4 acc[2][5];
    float bias[5];
    load_bias<5>(pk + kPConvB, 5 * (w >> 1), lane, bias);
    zero_tiles(acc);
    gemm_tiles<2, 5, kKConv / 16>(sX0, kLdX0, 2 * (w & 1), pk + kPConv, 5 * (w >> 1), lane, b_conv, acc);
    load_b0<1, kKLin / 16>(pk + kPLin, w >> 1, lane, b_lin);
    store_relu<2, 5>(sC1, kLdC1, 2 * (w & 1), 5 * (w >> 1), bias, lane, acc);
  }
  __syncthreads();
  CM3_STAMP(3, false);
  // ---- conv_linear : C1 [64][160] -> X2[:, 0:32], relu ---------------------------------------------------------------------
  {
    f32x4 acc[2][1];
    float bias[1];
    load_bias<1>(pk + kPLinB, w >> 1, lane, bias);
    zero_tiles(acc);
    gemm_tiles<2, 1, kKLin / 16>(sC1, kLdC1, 2 * (w & 1), pk + kPLin, w >> 1, lane, b_lin, acc);
    load_b0<4, kKSelf / 16>(pk + kPSelf, 4 * w, lane, b_self);
    store_relu<2, 1>(sX2, kLdX2, 2 * (w & 1), w >> 1, bias, lane, acc);
  }
  __syncthreads();
  CM3_STAMP(4, false);
  // ---- branch_self : X2 [64][48] -> H [64][256], relu; wave w owns columns [64w, 64w + 64) from here on ----------------------
  {
    f32x4 acc[4][4];
    float bias[4];
    load_bias<4>(pk + kPSelfB, 4 * w, lane, bias);
    zero_tiles(acc);
    gemm_tiles<4, 4, kKSelf / 16>(sX2, kLdX2, 0, pk + kPSelf, 4 * w, lane, b_self, acc);
    if constexpr (BF16) {
      load_b0_bf16<4>(pk + kPH2Sb, 4 * w, lane, b_h2b);
      store_relu_bf16<4, 4>(sHb, kLdHb, 0, 4 * w, bias, lane, acc);
    } else {
      load_b0<4, kH1 / 16>(pk + kPH2S, 4 * w, lane, b_h2);
      store_relu<4, 4>(sH, kLdH, 0, 4 * w, bias, lane, acc);
    }
  }
  __syncthreads();
  CM3_STAMP(5, false);
  // ---- h2 = relu(branch_self W_self_h2 + branch_others W_others_h2 + b) -------------------------------------------------------
  f32x4 acc2[4][4];
  zero_tiles(acc2);
  if constexpr (BF16) gemm_tiles_bf16<4>(sHb, kLdHb, pk + kPH2Sb, 4 * w, lane, b_h2b, acc2);
  else gemm_tiles<4, 4, kH1 / 16>(sH, kLdH, 0, pk + kPH2S, 4 * w, lane, b_h2, acc2);
  const bool stage2 = p.stage > 1;
  float bias_oth[4], bias_h2[4];
  load_bias<4>(pk + kPOthB, 4 * w, lane, bias_oth);
  load_bias<4>(pk + kPH2B, 4 * w, lane, bias_h2);
  if (stage2) load_b0<4, 1>(pk + kPOth, 4 * w, lane, b_oth);
  else load_b0<1, kH2 / 16>(pk + kPOut, 0, lane, b_out);
  CM3_STAMP(6, true);
  __syncthreads();  // every wave is done reading branch_self
  CM3_STAMP(7, false);
  if (stage2) {
    {  // branch_others: K = Lo <= 14 real inputs, ceil(Lo / 4) k-steps (see the staging of sXO)
      f32x4 acc[4][4];
      zero_tiles(acc);
      const int col = lane & 15, hi = lane >> 4, steps = (p.Lo + 3) >> 2;
      float4 a[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const float4 *>(&sXO[(16 * t + col) * kLdXO + 4 * hi]);
      for (int j = 0; j < steps; ++j) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float av = j == 0 ? a[t].x : (j == 1 ? a[t].y : (j == 2 ? a[t].z : a[t].w));
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float bv = j == 0 ? b_oth[c].x : (j == 1 ? b_oth[c].y : (j == 2 ? b_oth[c].z : b_oth[c].w));
            acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t][c], 0, 0, 0);
          }
        }
      }
      if constexpr (BF16) {
        load_b0_bf16<4>(pk + kPH2Ob, 4 * w, lane, b_h2b);
        store_relu_bf16<4, 4>(sHb, kLdHb, 0, 4 * w, bias_oth, lane, acc);
      } else {
        load_b0<4, kH1 / 16>(pk + kPH2O, 4 * w, lane, b_h2);
        store_relu<4, 4>(sH, kLdH, 0, 4 * w, bias_oth, lane, acc);
      }
    }
    __syncthreads();
    CM3_STAMP(8, false);
    if constexpr (BF16) gemm_tiles_bf16<4>(sHb, kLdHb, pk + kPH2Ob, 4 * w, lane, b_h2b, acc2);
    else gemm_tiles<4, 4, kH1 / 16>(sH, kLdH, 0, pk + kPH2O, 4 * w, lane, b_h2, acc2);
    load_b0<1, kH2 / 16>(pk + kPOut, 0, lane, b_out);
    CM3_STAMP(9, true);
    __syncthreads();
  }
  CM3_STAMP(10, false);
  store_relu<4, 4>(sH, kLdH, 0, 4 * w, bias_h2, lane, acc2);
  __syncthreads();
  CM3_STAMP(11, false);
  // ---- actor_out : wave w finishes rows [16w, 16w + 16) -------------------------------------------------------------------------
  {
    f32x4 acc[1][1];
    zero_tiles(acc);
    gemm_tiles<1, 1, kH2 / 16>(sH, kLdH, w, pk + kPOut, 0, lane, b_out, acc);
    const int col = lane & 15, hi = lane >> 4;
    if (col < 8) {
      const float b = pk[kPOutB + col];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) sLG[16 * w + 4 * hi + reg][col] = acc[0][0][reg] + b;
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  ck_actor_head(p, sLG, w, lane, row_base, rows);
  CM3_STAMP(12, true);
}

// ---- the precision = 2 kernel: EVERY layer in split float16 ---------------------------------------------------------------------
// The five small layers are 14 % of the MACs but, on the exact-f32 MFMA, most of the matrix-core time that is left once the two
// 256 x 256 layers run in float16 (measured, 16 384 rows: 256 x 256 layers only 30.4 us per launch, all layers 27.1, this kernel
// 23.3; every step in profiles/r03_checkers_actor_split_precision.txt).  Same tile loop for every layer: A from float16 hi / lo LDS planes, B from the packed hi / lo
// tiles, three MFMAs per (row tile, column tile, k-step of 32) -- two where the activations are exact in float16 (the window
// bytes are -1 / 0 / 1: no lo plane).
// (probe builds only, tools/r6: -DCM3_PROBE_B_L1 makes every weight request of the split-float16 kernel hit ONE 1 KB block -- an L1
// hit -- to tell the L2 -> L1 delivery of the 770 KB of weights apart from everything else; never defined in the product build)
#ifdef CM3_PROBE_B_L1
__device__ int cm3_probe_zero = 0;   // (a run-time zero: a literal one lets the compiler merge the MFMAs of column tiles that now read the same weights)
#define CM3_PROBE_BIDX(x) ((x) * cm3_probe_zero)
#else
#define CM3_PROBE_BIDX(x) (x)
#endif
#ifndef CM3_CK_A_BUFS
#define CM3_CK_A_BUFS 1
#endif
#ifndef CM3_CK_SELF_SPLIT
#define CM3_CK_SELF_SPLIT 1
#endif
template <int CT, int KS>
__device__ __forceinline__ void load_bx(const float *Bh, const float *Bl, int ct0, int lane, uint4 (&b0)[2][CT]) {
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    b0[0][c] = (reinterpret_cast<const uint4 *>(Bh) + CM3_PROBE_BIDX((size_t)(ct0 + c) * KS) * 64 + lane)[0];
    b0[1][c] = (reinterpret_cast<const uint4 *>(Bl) + CM3_PROBE_BIDX((size_t)(ct0 + c) * KS) * 64 + lane)[0];
  }
}

// TRANSPOSED tiles: the weights are the A operand and the activations the B operand, so a lane ends with FOUR CONSECUTIVE UNITS
// (4 (l >> 4) + reg of column tile c) of ONE agent row (16 (rt0 + t) + (l & 15)) -- contiguous in the next layer's row-major
// plane: one 8-byte LDS store per plane and tile instead of four 2-byte stores (the epilogues were a quarter of this kernel).
// Both operand fragments are "8 consecutive k of index l & 15" in the same registers as before; only their roles swap.
// SWZ: the activation planes written by store_relu_x3b<.., true> -- 16-byte chunk q of a row sits at q ^ ((row >> 2) & 1), see there.
template <int RT, int CT, int KS, bool ALO, bool SWZ = false>
__device__ __forceinline__ void gemm_x3(const _Float16 *Ah, const _Float16 *Al, int lda, int rt0, const float *Bh, const float *Bl,
                                        int ct0, int lane, const uint4 (&b0)[2][CT], f32x4 (&acc)[RT][CT]) {
  const int col = lane & 15, hi = SWZ ? ((lane >> 4) ^ ((lane >> 2) & 1)) : (lane >> 4);
  const uint4 *bsrc[2][CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    bsrc[0][c] = reinterpret_cast<const uint4 *>(Bh) + CM3_PROBE_BIDX((size_t)(ct0 + c) * KS) * 64 + lane;
    bsrc[1][c] = reinterpret_cast<const uint4 *>(Bl) + CM3_PROBE_BIDX((size_t)(ct0 + c) * KS) * 64 + lane;
  }
  // weights of k-step st + 2 are requested before the MFMAs of step st issue (a ring of three: one step of MFMAs, 768 cycles for
  // the 4 x 4 tiles, is shorter than the L2 round trip of a lone workgroup)
  uint4 bq[3][2][CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    bq[0][0][c] = b0[0][c];
    bq[0][1][c] = b0[1][c];
    if (KS > 1) {
      bq[1][0][c] = bsrc[0][c][CM3_PROBE_BIDX(64)];
      bq[1][1][c] = bsrc[1][c][CM3_PROBE_BIDX(64)];
    }
  }
  // All activation fragments of a k-step are requested BEFORE its matrix instructions, behind a scheduling barrier (round 6, late: left to
  // itself the compiler sinks every LDS read to its first use -- read, wait, two matrix instructions, read, wait, ... -- and the 256-deep
  // h2 pass ran at 52 % of its matrix time).  CM3_CK_A_BUFS = 2 requests step st + 1's fragments ahead of step st's instructions (two
  // register sets): measured twice, with 24 and with 10 spilled registers, the same time both times -- the default stays 1.
  constexpr int NB = CM3_CK_A_BUFS;   // 2: fragments of step st + 1 requested before the matrix instructions of step st; 1: of step st, all at once
  f16x8 ah[NB][RT], al[NB][RT];
  auto read_a = [&](int st, int buf) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      ah[buf][t] = *reinterpret_cast<const f16x8 *>(Ah + (16 * (rt0 + t) + col) * lda + 32 * st + 8 * hi);
      if constexpr (ALO) al[buf][t] = *reinterpret_cast<const f16x8 *>(Al + (16 * (rt0 + t) + col) * lda + 32 * st + 8 * hi);
    }
  };
  if (NB == 2) read_a(0, 0);
#pragma unroll
  for (int st = 0; st < KS; ++st) {
    if (st + 2 < KS) {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        bq[(st + 2) % 3][0][c] = bsrc[0][c][CM3_PROBE_BIDX((st + 2) * 64)];
        bq[(st + 2) % 3][1][c] = bsrc[1][c][CM3_PROBE_BIDX((st + 2) * 64)];
      }
    }
    if (NB == 1) {
      read_a(st, 0);
      if (KS > 1) __builtin_amdgcn_sched_barrier(0);
    } else if (KS > 1) {
      if (st + 1 < KS) read_a(st + 1, (st + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int cur = NB == 2 ? (st & 1) : 0;
    // the three products one after the other over ALL tiles: RT x CT independent accumulators between two MFMAs on the same one
    f16x8 wh[CT], wl[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      __builtin_memcpy(&wh[c], &bq[st % 3][0][c], 16);
      __builtin_memcpy(&wl[c], &bq[st % 3][1][c], 16);
    }
    if constexpr (ALO) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], al[cur][t], acc[t][c], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c], ah[cur][t], acc[t][c], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], ah[cur][t], acc[t][c], 0, 0, 0);
    if (KS > 1) __builtin_amdgcn_sched_barrier(0);
  }
}

// ALL weights of a short layer (KS <= 5 k-steps) at once, requested by the caller BEFORE the previous layer's epilogue and barrier
// (round 6): the small layers have 15-48 matrix instructions and could not hide the L2 round trips of a ring -- conv_linear's five
// k-steps were three exposed round trips (~2.5 k cycles for 240 cycles of matrix work).  Same products in the same order as gemm_x3.
template <int CT, int KS>
__device__ __forceinline__ void load_bx_all(const float *Bh, const float *Bl, int ct0, int lane, uint4 (&ball)[KS][2][CT]) {
#pragma unroll
  for (int st = 0; st < KS; ++st)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      ball[st][0][c] = (reinterpret_cast<const uint4 *>(Bh) + CM3_PROBE_BIDX(((size_t)(ct0 + c) * KS + st)) * 64 + lane)[0];
      ball[st][1][c] = (reinterpret_cast<const uint4 *>(Bl) + CM3_PROBE_BIDX(((size_t)(ct0 + c) * KS + st)) * 64 + lane)[0];
    }
}

template <int RT, int CT, int KS, bool ALO, bool SWZ = false>
__device__ __forceinline__ void gemm_x3_pre(const _Float16 *Ah, const _Float16 *Al, int lda, int rt0, int lane,
                                            const uint4 (&ball)[KS][2][CT], f32x4 (&acc)[RT][CT]) {
  const int col = lane & 15, hi = SWZ ? ((lane >> 4) ^ ((lane >> 2) & 1)) : (lane >> 4);
  // every activation fragment of the layer first, ONE wait, then the matrix instructions (see gemm_x3: the compiler's own order is
  // read, wait, two instructions, read, wait, ...)
  f16x8 ah[KS][RT], al[KS][RT];
#pragma unroll
  for (int st = 0; st < KS; ++st)
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      ah[st][t] = *reinterpret_cast<const f16x8 *>(Ah + (16 * (rt0 + t) + col) * lda + 32 * st + 8 * hi);
      if constexpr (ALO) al[st][t] = *reinterpret_cast<const f16x8 *>(Al + (16 * (rt0 + t) + col) * lda + 32 * st + 8 * hi);
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int st = 0; st < KS; ++st) {
    f16x8 wh[CT], wl[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      __builtin_memcpy(&wh[c], &ball[st][0][c], 16);
      __builtin_memcpy(&wl[c], &ball[st][1][c], 16);
    }
    if constexpr (ALO) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], al[st][t], acc[t][c], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c], ah[st][t], acc[t][c], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c], ah[st][t], acc[t][c], 0, 0, 0);
  }
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// this lane's four bias values per column tile (units 16 (ct0 + c) + 4 (l >> 4) + reg)
template <int CT> __device__ __forceinline__ void load_bias4(const float *bias, int ct0, int lane, float4 (&b)[CT]) {
#pragma unroll
  for (int c = 0; c < CT; ++c) b[c] = *reinterpret_cast<const float4 *>(bias + 16 * (ct0 + c) + 4 * (lane >> 4));
}

// The bias rides in as the accumulators' start value (round 6: one VALU instruction per value less in every epilogue; a unit's bias
// is then the FIRST term of its sum instead of the last): a lane's four units of column tile c are the same for every row tile
template <int RT, int CT> __device__ __forceinline__ void bias_tiles(const float4 (&bias)[CT], f32x4 (&acc)[RT][CT]) {
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[t][c] = f32x4{bias[c].x, bias[c].y, bias[c].z, bias[c].w};
}

// v = hi + lo + O(2^-22 v): hi = float16(v), lo = float16(v - hi), two values per conversion instruction (v_cvt_pk_f16_f32; the
// subtraction stays scalar: packed float32 arithmetic next to matrix instructions is an anti-lever -- and the family the lint watches)
__device__ __forceinline__ void split4(const float (&v)[4], f16x4 &vh, f16x4 &vl) {
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const f2v x = {v[2 * p], v[2 * p + 1]};
    const h2v hp = __builtin_convertvector(x, h2v);
    const float r0 = x[0] - (float)hp[0], r1 = x[1] - (float)hp[1];
    const h2v lp = __builtin_convertvector(f2v{r0, r1}, h2v);
    vh[2 * p] = hp[0]; vh[2 * p + 1] = hp[1];
    vl[2 * p] = lp[0]; vl[2 * p + 1] = lp[1];
  }
}

// O[agent row][unit] = relu(acc) as float16 hi / lo planes, transposed tiles (see gemm_x3); the bias is already in acc (bias_tiles).
// SWZ: lanes 0..15 of an 8-byte store are rows 0..15 at ONE column, and a row stride that keeps the 16-byte reads conflict-free
// (8 x odd dwords) puts rows r, r + 4, r + 8, r + 12 on the same banks -- a 4-way conflict on every store of every epilogue (224 per
// workgroup and tick, most of the kernel's SQ_LDS_BANK_CONFLICT).  With the 16-byte chunks of rows 4..7 and 12..15 swapped in pairs
// (chunk q at q ^ ((row >> 2) & 1)) the stores are 2-way and the reads stay conflict-free; a reader flips the same bit (gemm_x3).
template <int RT, int CT, bool SWZ = false>
__device__ __forceinline__ void store_relu_x3b(_Float16 *Oh, _Float16 *Ol, int ldo, int rt0, int ct0, int lane, const f32x4 (&acc)[RT][CT]) {
  const int col = lane & 15, hi = SWZ ? ((lane >> 4) ^ ((lane >> 1) & 2)) : (lane >> 4);
#pragma unroll
  for (int c = 0; c < CT; ++c) {
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      // (relu_f32: ONE integer maximum on the bit pattern; fmaxf is two instructions -- it quiets a NaN first)
      const float v[4] = {relu_f32(acc[t][c][0]), relu_f32(acc[t][c][1]), relu_f32(acc[t][c][2]), relu_f32(acc[t][c][3])};
      f16x4 vh, vl;
      split4(v, vh, vl);
      const int at = (16 * (rt0 + t) + col) * ldo + 16 * (ct0 + c) + 4 * hi;
      *reinterpret_cast<f16x4 *>(Oh + at) = vh;
      *reinterpret_cast<f16x4 *>(Ol + at) = vl;
    }
  }
}

// O[agent row][unit] = relu(acc + bias[unit]) as float16 hi / lo planes, transposed tiles (see gemm_x3) -- the f32 / bf16 kernels' form
template <int RT, int CT>
__device__ __forceinline__ void store_relu_x3(_Float16 *Oh, _Float16 *Ol, int ldo, int rt0, int ct0, const float4 (&bias)[CT], int lane,
                                              const f32x4 (&acc)[RT][CT]) {
  const int col = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const float bs[4] = {bias[c].x, bias[c].y, bias[c].z, bias[c].w};
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      f16x4 vh, vl;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float v = fmaxf(acc[t][c][reg] + bs[reg], 0.0f);
        vh[reg] = (_Float16)v;
        vl[reg] = (_Float16)(v - (float)vh[reg]);
      }
      const int at = (16 * (rt0 + t) + col) * ldo + 16 * (ct0 + c) + 4 * hi;
      *reinterpret_cast<f16x4 *>(Oh + at) = vh;
      *reinterpret_cast<f16x4 *>(Ol + at) = vl;
    }
  }
}

__device__ __forceinline__ void put_split(_Float16 *h, _Float16 *l, int at, float v) {
  const _Float16 vh = (_Float16)v;
  h[at] = vh;
  l[at] = (_Float16)(v - (float)vh);
}

// LDS of the split-float16 forward pass: float16 hi | lo planes.  H [64][264] holds, one after the other, branch_others, C1 (the
// conv's output, [64][168]), branch_self and h2; X0 (the window bytes, exact in float16: hi only), X2 (conv_linear | tail) and XO
// (v_obs_others) have storage of their own.
struct CkX3Planes {
  _Float16 *Hh, *Hl, *X0, *C1h, *C1l, *X2h, *X2l, *XOh, *XOl;
  float (*LG)[8];
};
constexpr int kCkX3HBytes = 2 * 64 * ck_actor::kLdHb * 2, kCkX3X0Bytes = 64 * ck_actor::kLhX0 * 2,
              kCkX3X2Bytes = 2 * 64 * ck_actor::kLhX2 * 2, kCkX3XOBytes = 2 * 64 * ck_actor::kLhXO * 2;
static_assert(2 * 64 * ck_actor::kLhC1 * 2 <= kCkX3HBytes, "the C1 planes must fit into the H storage");

// ---- the forward pass of a 512-thread workgroup (8 waves: two per SIMD) that owns 64 agent rows --------------------------------------
// Round 3's kernel ran 4 waves, one per SIMD, alone on its CU (98 KB of LDS): 16 k of its 49.5 k cycles on the matrix cores.  With 8
// waves every layer's output tiles split eight ways (a wave's MFMAs, LDS reads and weight loads halve, the SIMD's other wave fills its
// waits); each column tile still belongs to ONE wave, so the weights cross the CU's L1 once per workgroup.  Measured, same box: 23.5 ->
// 22.7 us per launch at 16 384 rows, same bits (profiles/r06_checkers_policy.txt) -- the h2 passes are MFMA-bound now (19 cycles per
// matrix instruction per SIMD), the rest of the launch is the serial chain of small layers.
//   tiles (16 x 16) per layer:   conv 4 x 10   lin 4 x 2   branch_self / branch_others / h2 4 x 16   out 4 x 1
//   per wave:                    conv 4 x 1 + 1 x 1   lin 1 x 1   4 x 2 (all rows, 32 units)       out: every wave its 32 units
// ORDER (round 6): the others branch goes FIRST -- h2's accumulators start with branch_others W_others_h2 (k ascending) and the
// branch_self terms follow -- so that what the others branch contributes is the accumulators' state after ck_x3_others(): a pure
// function of a row's v_obs_others.  The whole-episode kernel (policy_checkers.hip) reads that state from a table instead
// (cm3_actor_checkers_pack builds it with this very function over the 91 cells another agent can stand on), bit for bit.
constexpr int kCkBCT = 2;   // column tiles per wave in the 256-wide layers
// The conv's 4 x 10 output tiles over eight waves: wave w takes column tile w for ALL four row tiles and one tile of the last two
// column tiles (column tile 8 + (w >> 2), row tile w & 3).  (First cut of round 6: row tile w & 3 x five column tiles -- four waves then
// loaded the SAME 30 KB of weights each, 240 KB per workgroup and tick through a 64 B / clock L1 path for 61 KB of distinct weights; a
// what-if build without the three small layers showed them costing 6 of a tick's 15 us for 0.7 us of matrix work.  Now 96 KB.)
typedef uint4 CkConvB[ck_actor::kKConvX / 32][2][2];   // [k-step][hi | lo plane][own column tile | the shared one]
__device__ __forceinline__ void load_bx_all_into(const float *Bh, const float *Bl, int ct, int lane, int slot, CkConvB &b) {
  constexpr int KS = ck_actor::kKConvX / 32;
#pragma unroll
  for (int st = 0; st < KS; ++st) {
    b[st][0][slot] = (reinterpret_cast<const uint4 *>(Bh) + CM3_PROBE_BIDX(((size_t)ct * KS + st)) * 64 + lane)[0];
    b[st][1][slot] = (reinterpret_cast<const uint4 *>(Bl) + CM3_PROBE_BIDX(((size_t)ct * KS + st)) * 64 + lane)[0];
  }
}
// ALL of the conv's weights of wave w (12 sixteen-byte loads)
__device__ __forceinline__ void ck_x3_load_conv(const float *pk, int w, int lane, CkConvB &b) {
  load_bx_all_into(pk + ck_actor::kXConvH, pk + ck_actor::kXConvL, w, lane, 0, b);
  load_bx_all_into(pk + ck_actor::kXConvH, pk + ck_actor::kXConvL, 8 + (w >> 2), lane, 1, b);
}
#ifndef CM3_X3_CONV_STAMP
#define CM3_X3_CONV_STAMP 3   // (probe builds: the timeline slot of "conv done"; the policy probe moves it off ck_tick_env's slot 3)
#endif

// h2's accumulators start from its bias b (wave w: units [32w, 32w + 32))
__device__ __forceinline__ void ck_x3_h2_bias(const float *pk, int w, int lane, f32x4 (&acc2)[4][kCkBCT]) {
  float4 bias_h2[kCkBCT];
  load_bias4<kCkBCT>(pk + ck_actor::kPH2B, kCkBCT * w, lane, bias_h2);
  bias_tiles(bias_h2, acc2);
}

// h2 accumulators += branch_others W_others_h2.  Reads XO, uses the H planes.  Enter with XO visible to the workgroup; leaves BEHIND a
// barrier: the H planes are free.
__device__ __forceinline__ void ck_x3_others(const CkX3Planes &L, const float *pk, int w, int lane, f32x4 (&acc2)[4][kCkBCT]) {
  using namespace ck_actor;
  constexpr int BCT = kCkBCT;
  uint4 b_oth[2][BCT], b_h2[2][BCT];
  float4 bias_oth[BCT];
  load_bx<BCT, 1>(pk + kXOthH, pk + kXOthL, BCT * w, lane, b_oth);
  load_bias4<BCT>(pk + kPOthB, BCT * w, lane, bias_oth);
  {
    f32x4 acc[4][BCT];
    bias_tiles(bias_oth, acc);
    gemm_x3<4, BCT, 1, true>(L.XOh, L.XOl, kLhXO, 0, pk + kXOthH, pk + kXOthL, BCT * w, lane, b_oth, acc);
    load_bx<BCT, 8>(pk + kPH2Oh, pk + kPH2Ol, BCT * w, lane, b_h2);
    store_relu_x3b<4, BCT, true>(L.Hh, L.Hl, kLdHb, 0, BCT * w, lane, acc);
  }
  __syncthreads();
  CM3_STAMP(8, false);
  gemm_x3<4, BCT, 8, true, true>(L.Hh, L.Hl, kLdHb, 0, pk + kPH2Oh, pk + kPH2Ol, BCT * w, lane, b_h2, acc2);
  CM3_STAMP(9, true);
  __syncthreads();
}

// conv -> conv_linear -> branch_self -> h2 (+= on acc2) -> actor_out.  Enter with X0 and the tail of X2 visible and the H planes free;
// b_conv: ALL of the conv's weights (load_bx_all), requested by the caller ahead of time.  Leaves the workgroup BEHIND its last barrier with the
// logits of rows [16w, 16w + 16) written by wave w < 4 (not yet visible to other waves).
// HOOKS: the whole-episode kernel starts acc2 from the others-branch table and fetches those rows along the way -- after_conv()
// (behind the conv's barrier: X0 is dead, the conv's 120 registers of weights are gone), after_lin() (behind conv_linear's barrier)
// and before_h2(acc2) (behind branch_self's barrier, ahead of the h2 pass); the stand-alone kernel passes none.
struct CkNoHooks {
  __device__ __forceinline__ void mid_conv() const {}
  __device__ __forceinline__ void after_conv() const {}
  __device__ __forceinline__ void after_lin() const {}
  __device__ __forceinline__ void before_h2(f32x4 (&)[4][kCkBCT]) const {}
};
template <class HOOKS = CkNoHooks>
__device__ __forceinline__ void ck_x3_self_chain(const CkX3Planes &L, const float *pk, int w, int lane, const CkConvB &b_conv,
                                                 f32x4 (&acc2)[4][kCkBCT], HOOKS hooks = HOOKS()) {
  using namespace ck_actor;
  constexpr int BCT = kCkBCT;
  const int s_rt0 = w & 3, s_half = w >> 2;
  uint4 b_lin[kKLin / 32][2][1], b_self[kKSelfX / 32][2][BCT], b_h2[2][BCT];
  // ---- conv (Toeplitz): X0 [64][96] -> C1 [64][160], relu ------------------------------------------------------------------------
#ifndef CM3_PROBE_SKIP_SMALL   // (probe builds only: without conv / conv_linear / branch_self)
#ifndef CM3_PROBE_SKIP_CONV
  {
    constexpr int KS = kKConvX / 32;
    uint4 b_own[KS][2][1], b_sh[KS][2][1];
#pragma unroll
    for (int st = 0; st < KS; ++st)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        b_own[st][pl][0] = b_conv[st][pl][0];
        b_sh[st][pl][0] = b_conv[st][pl][1];
      }
    f32x4 acc[4][1], acc_sh[1][1];
    float4 bias[1], bias_sh[1];
    load_bias4<1>(pk + kPConvB, w, lane, bias);
    load_bias4<1>(pk + kPConvB, 8 + s_half, lane, bias_sh);
    bias_tiles(bias, acc);
    bias_tiles(bias_sh, acc_sh);
    gemm_x3_pre<4, 1, KS, false>(L.X0, L.X0, kLhX0, 0, lane, b_own, acc);
    gemm_x3_pre<1, 1, KS, false>(L.X0, L.X0, kLhX0, s_rt0, lane, b_sh, acc_sh);
    hooks.mid_conv();
    load_bx_all<1, kKLin / 32>(pk + kXLinH, pk + kXLinL, s_half, lane, b_lin);
    store_relu_x3b<4, 1, true>(L.C1h, L.C1l, kLhC1, 0, w, lane, acc);
    store_relu_x3b<1, 1, true>(L.C1h, L.C1l, kLhC1, s_rt0, 8 + s_half, lane, acc_sh);
  }
#else
  load_bx_all<1, kKLin / 32>(pk + kXLinH, pk + kXLinL, s_half, lane, b_lin);
#endif
  __syncthreads();
  CM3_STAMP(CM3_X3_CONV_STAMP, false);
  hooks.after_conv();
  // ---- conv_linear: C1 [64][160] -> X2[:, 0:32], relu ---------------------------------------------------------------------------
#ifndef CM3_PROBE_SKIP_LIN
  {
    f32x4 acc[1][1];
    float4 bias[1];
    load_bias4<1>(pk + kPLinB, s_half, lane, bias);
    bias_tiles(bias, acc);
    gemm_x3_pre<1, 1, kKLin / 32, true, true>(L.C1h, L.C1l, kLhC1, s_rt0, lane, b_lin, acc);
    load_bx_all<BCT, kKSelfX / 32>(pk + kXSelfH, pk + kXSelfL, BCT * w, lane, b_self);
    store_relu_x3b<1, 1>(L.X2h, L.X2l, kLhX2, s_rt0, s_half, lane, acc);
  }
#else
  load_bx_all<BCT, kKSelfX / 32>(pk + kXSelfH, pk + kXSelfL, BCT * w, lane, b_self);
#endif
  __syncthreads();
  CM3_STAMP(4, false);
  hooks.after_lin();
  // ---- branch_self: X2 [64][64] -> H [64][256], relu; wave w owns units [32w, 32w + 32) from here on ---------------------------------
#ifndef CM3_PROBE_SKIP_SELF
  {
#if CM3_CK_SELF_SPLIT
    // The two column tiles one after the other, the first one's epilogue (relu, float16 split, stores: ~70 vector instructions)
    // issued BETWEEN the second one's matrix instructions: both waves of a SIMD otherwise run their 48 matrix instructions and then
    // their ~140 vector instructions at the same time, one pipe idle in each half.  Same products in the same order per accumulator.
    static_assert(BCT == 2, "two column tiles per wave");
    constexpr int KS = kKSelfX / 32;
    f32x4 acc0[4][1], acc1[4][1];
    float4 bias[BCT];
    load_bias4<BCT>(pk + kPSelfB, BCT * w, lane, bias);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc0[t][0] = f32x4{bias[0].x, bias[0].y, bias[0].z, bias[0].w};
      acc1[t][0] = f32x4{bias[1].x, bias[1].y, bias[1].z, bias[1].w};
    }
    const int col = lane & 15, hi = lane >> 4;
    f16x8 ah[KS][4], al[KS][4];
#pragma unroll
    for (int st = 0; st < KS; ++st)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[st][t] = *reinterpret_cast<const f16x8 *>(L.X2h + (16 * t + col) * kLhX2 + 32 * st + 8 * hi);
        al[st][t] = *reinterpret_cast<const f16x8 *>(L.X2l + (16 * t + col) * kLhX2 + 32 * st + 8 * hi);
      }
    __builtin_amdgcn_sched_barrier(0);
    auto tile_col = [&](int c, f32x4 (&acc)[4][1]) {
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        f16x8 wh, wl;
        __builtin_memcpy(&wh, &b_self[st][0][c], 16);
        __builtin_memcpy(&wl, &b_self[st][1][c], 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[st][t], acc[t][0], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[st][t], acc[t][0], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[st][t], acc[t][0], 0, 0, 0);
      }
    };
    tile_col(0, acc0);
    load_bx<BCT, 8>(pk + kPH2Sh, pk + kPH2Sl, BCT * w, lane, b_h2);
    // second tile: matrix instruction k, then step k of the first tile's epilogue (store_relu_x3b<.., true>'s operations, six steps per
    // row tile), pinned by a scheduling barrier each (asked for the same order through sched_group_barrier the scheduler gave up
    // after three groups)
    {
      typedef _Float16 h2v __attribute__((ext_vector_type(2)));
      typedef float f2v __attribute__((ext_vector_type(2)));
      const int hs = hi ^ ((lane >> 1) & 2);     // the swizzled 8-byte position (store_relu_x3b)
      float v[4][4];
      h2v hp[4][2], lp[4][2];
      float r[4][4];
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        const int st = k / 12, prod = (k / 4) % 3, t = k % 4;
        f16x8 wh, wl;
        __builtin_memcpy(&wh, &b_self[st][0][1], 16);
        __builtin_memcpy(&wl, &b_self[st][1][1], 16);
        acc1[t][0] = prod == 0   ? __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[st][t], acc1[t][0], 0, 0, 0)
                     : prod == 1 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[st][t], acc1[t][0], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[st][t], acc1[t][0], 0, 0, 0);
        const int et = k / 6, part = k % 6;
        if (part < 2) {
          v[et][2 * part] = relu_f32(acc0[et][0][2 * part]);
          v[et][2 * part + 1] = relu_f32(acc0[et][0][2 * part + 1]);
          hp[et][part] = __builtin_convertvector(f2v{v[et][2 * part], v[et][2 * part + 1]}, h2v);
        } else if (part < 4) {
          const int pp = part - 2;
          r[et][2 * pp] = v[et][2 * pp] - (float)hp[et][pp][0];
          r[et][2 * pp + 1] = v[et][2 * pp + 1] - (float)hp[et][pp][1];
        } else if (part == 4) {
          lp[et][0] = __builtin_convertvector(f2v{r[et][0], r[et][1]}, h2v);
          lp[et][1] = __builtin_convertvector(f2v{r[et][2], r[et][3]}, h2v);
        } else {
          const f16x4 vh = {hp[et][0][0], hp[et][0][1], hp[et][1][0], hp[et][1][1]};
          const f16x4 vl = {lp[et][0][0], lp[et][0][1], lp[et][1][0], lp[et][1][1]};
          const int at = (16 * et + col) * kLdHb + 16 * (BCT * w) + 4 * hs;
          *reinterpret_cast<f16x4 *>(L.Hh + at) = vh;
          *reinterpret_cast<f16x4 *>(L.Hl + at) = vl;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    store_relu_x3b<4, 1, true>(L.Hh, L.Hl, kLdHb, 0, BCT * w + 1, lane, acc1);
#else
    f32x4 acc[4][BCT];
    float4 bias[BCT];
    load_bias4<BCT>(pk + kPSelfB, BCT * w, lane, bias);
    bias_tiles(bias, acc);
    gemm_x3_pre<4, BCT, kKSelfX / 32, true>(L.X2h, L.X2l, kLhX2, 0, lane, b_self, acc);
    load_bx<BCT, 8>(pk + kPH2Sh, pk + kPH2Sl, BCT * w, lane, b_h2);
    store_relu_x3b<4, BCT, true>(L.Hh, L.Hl, kLdHb, 0, BCT * w, lane, acc);
#endif
  }
#else
  load_bx<BCT, 8>(pk + kPH2Sh, pk + kPH2Sl, BCT * w, lane, b_h2);
#endif
  __syncthreads();
  CM3_STAMP(5, false);
#else
  hooks.after_conv();
  hooks.after_lin();
  load_bx<BCT, 8>(pk + kPH2Sh, pk + kPH2Sl, BCT * w, lane, b_h2);
  __syncthreads();
#endif
  hooks.before_h2(acc2);
  // ---- h2 = relu(b + branch_others W_others_h2 [both already in acc2] + branch_self W_self_h2) ------------------------------------
#ifdef CM3_PROBE_H2_KS      // (probe builds only: the h2 pass with fewer k-steps)
  gemm_x3<4, BCT, CM3_PROBE_H2_KS, true, true>(L.Hh, L.Hl, kLdHb, 0, pk + kPH2Sh, pk + kPH2Sl, BCT * w, lane, b_h2, acc2);
#else
  gemm_x3<4, BCT, 8, true, true>(L.Hh, L.Hl, kLdHb, 0, pk + kPH2Sh, pk + kPH2Sl, BCT * w, lane, b_h2, acc2);
#endif
  // ---- actor_out, round 6: every wave contracts ITS 32 units of h2 straight from the accumulator registers -- a lane's eight values of
  // an agent row (two column tiles x four units) are the eight consecutive k of one float16 matrix instruction's B operand once the
  // output weights are packed in that unit order (k_ck_actor_pack, layer 6) -- and leaves a partial logit per agent row in LDS; the
  // row's wave adds the eight partials in wave order.  (Until round 6: relu(h2) -> LDS as float16 planes, barrier, waves 0..3 ran the
  // 256-deep layer while waves 4..7 waited: 4.1 k cycles of a 37 k tick.)
  uint4 wo[2];
  wo[0] = (reinterpret_cast<const uint4 *>(pk + kXOutH) + (size_t)w * 64 + lane)[0];
  wo[1] = (reinterpret_cast<const uint4 *>(pk + kXOutL) + (size_t)w * 64 + lane)[0];
  f16x8 woh, wol;
  __builtin_memcpy(&woh, &wo[0], 16);
  __builtin_memcpy(&wol, &wo[1], 16);
  f32x4 po[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f16x4 h0, l0, h1, l1;
    const float v0[4] = {relu_f32(acc2[t][0][0]), relu_f32(acc2[t][0][1]), relu_f32(acc2[t][0][2]), relu_f32(acc2[t][0][3])};
    const float v1[4] = {relu_f32(acc2[t][1][0]), relu_f32(acc2[t][1][1]), relu_f32(acc2[t][1][2]), relu_f32(acc2[t][1][3])};
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    const f16x8 ah = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    const f16x8 al = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(woh, al, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wol, ah, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(woh, ah, acc, 0, 0, 0);
    po[t] = acc;
  }
  CM3_STAMP(6, true);
#ifndef CM3_PROBE_NO_TAIL_BARRIERS   // (probe builds only: what the three barriers between the h2 pass and the env phase cost)
  __syncthreads();  // every wave is done reading branch_self: the H storage takes the partial logits, float [8 waves][64 rows][8]
#endif
  CM3_STAMP(10, false);
  float *part = reinterpret_cast<float *>(L.Hh);
  {
    const int col = lane & 15, hi = lane >> 4;
    if (hi < 2) {   // actions 4 hi .. 4 hi + 3 of agent row 16 t + col (actions 5 .. 7: zero weights)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        *reinterpret_cast<float4 *>(part + ((size_t)w * 64 + 16 * t + col) * 8 + 4 * hi) = make_float4(po[t][0], po[t][1], po[t][2], po[t][3]);
    }
  }
#ifndef CM3_PROBE_NO_TAIL_BARRIERS
  __syncthreads();
#endif
  CM3_STAMP(11, false);
  if (w < 4 && lane < 16) {   // logits of agent row 16 w + lane: b_out + the eight partials in wave order
    float lg[kA];
#pragma unroll
    for (int a = 0; a < kA; ++a) lg[a] = pk[kPOutB + a];
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) {
      const float4 p4 = *reinterpret_cast<const float4 *>(part + ((size_t)ww * 64 + 16 * w + lane) * 8);
      const float p5 = part[((size_t)ww * 64 + 16 * w + lane) * 8 + 4];
      lg[0] += p4.x; lg[1] += p4.y; lg[2] += p4.z; lg[3] += p4.w; lg[4] += p5;
    }
#pragma unroll
    for (int a = 0; a < kA; ++a) L.LG[16 * w + lane][a] = lg[a];
  }
}

// Staging of the stand-alone actor kernel's inputs from the env's output buffers, by threads tid < 256 (four lanes per agent row):
// every global load is issued first, the zero fills run while they are in flight, then the scatter.
__device__ __forceinline__ void ck_x3_stage_inputs(const CkActorParams &p, const CkX3Planes &L, int tid, size_t row_base, size_t rows) {
  using namespace ck_actor;
  const int N = p.N;
  _Float16 *sX0 = L.X0, *sX2h = L.X2h, *sX2l = L.X2l, *sXOh = L.XOh, *sXOl = L.XOl;
  // ---- stage the inputs: every global load is issued first, the zero fills run while they are in flight, then the scatter ----------
  // (a lone workgroup per CU hides nothing: staged one dependent load after the other this phase was a fifth of the launch)
  // fast path: N a power of two <= 64 (the 64 rows are whole envs), dword-aligned env records, and 4 N lanes per env cover a
  // record in kWin dwords each (true for records padded to a dword: 75 N bytes <= 80 N); anything else takes the byte path.
  // Lane li of env el loads dwords li, li + 4 N, ...: shifts only (a runtime division is ~40 instructions on this hardware).
  constexpr int kWin = 5;
  const int dpe = p.obst_stride >> 2, rec = N * kObs;
  const int lsh = 2 + (__ffs(N) - 1), lpe = 4 * N;   // lanes per env = 256 / (64 / N)
  const bool dwords = (N & (N - 1)) == 0 && N <= 64 && (p.obst_stride & 3) == 0 && dpe <= kWin * lpe;
  const int wel = tid >> lsh, wli = tid & (lpe - 1);
  uint32_t wv[kWin];
  if (dwords) {
    size_t e = row_base / N + wel;
    e = e < (size_t)p.E ? e : (size_t)p.E - 1;
    const uint32_t *recp = reinterpret_cast<const uint32_t *>(p.obs_self_t + e * (size_t)p.obst_stride);
#pragma unroll
    for (int q = 0; q < kWin; ++q) {
      const int dd = wli + lpe * q;
      wv[q] = recp[dd < dpe ? dd : 0];
    }
  }
  // the concat tail and v_obs_others, four lanes per agent row: v_obs_self | a_prev + goal one-hots | pad | v_obs_others
  const int trow = tid >> 2, part = tid & 3;
  size_t row_t = row_base + trow;
  row_t = row_t < rows ? row_t : rows - 1;
  // Every lane requests ALL of its row's tail (the four lanes of a row repeat each other's addresses: same cache lines) instead of
  // each part branching to its own loads.  The branches were executed one after the other, each with its own wait: prev_done ->
  // actions_prev (two dependent round trips), then v_obs_others in a loop of load / wait per value -- four memory round trips
  // behind each other in a phase that was a sixth of the launch (round 4, late; same disease as the particle actor's table copy).
  double tv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) tv[k] = p.obs_self_v[row_t * 4 + k];
  // v_obs_others: Lo = 2 (N - 1) <= 14 doubles per row (ck_actor_check), requested as pairs with a CLAMPED pair index -- straight-line
  // loads (a predicate per value compiled to a branch, a load and a wait each)
  constexpr int kLoMax = 14;
  typedef double ck_dbl2 __attribute__((ext_vector_type(2)));
  ck_dbl2 ov2[kLoMax / 2];
  {
    const int pairs = p.Lo >> 1;
    const double *orow = p.obs_others + row_t * p.Lo;
#pragma unroll
    for (int j = 0; j < kLoMax / 2; ++j) {
      const int jc = j < pairs ? j : (pairs > 0 ? pairs - 1 : 0);
      ck_dbl2 v;
      __builtin_memcpy(&v, orow + 2 * jc, 16);
      ov2[j] = v;
    }
  }
  // a fresh episode starts from actions_prev = zeros (train_onpolicy.py:295)
  const int prev_done_t = p.prev_done ? (int)p.prev_done[row_t / N] : 0;
  const int prev_act_t = p.actions_prev ? (int)p.actions_prev[row_t] : 0;
  const int gl = p.goals[row_t];
  // (no `ap = prev_done ? 0 : prev_act` here: a select with a loaded operand is turned back into a branch around the load, i.e.
  // into the dependent round trip this block removes -- the one-hot below tests both values instead)
  // zero fills: the k padding of X0, the tail of X2 (units 32 .. 63: values follow below, same wave, program order), the XO rows
  if (dwords) {
    for (int idx = tid; idx < 64 * (kKConvX - kObs); idx += 256) {
      const int r = idx / (kKConvX - kObs), k = kObs + idx - r * (kKConvX - kObs);
      sX0[r * kLhX0 + k] = (_Float16)0.0f;
    }
  }
  {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    if (part == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<uint4 *>(sX2h + trow * kLhX2 + kLin + 8 * q) = z;
        *reinterpret_cast<uint4 *>(sX2l + trow * kLhX2 + kLin + 8 * q) = z;
      }
    } else if (part == 3) {
#pragma unroll
      for (int q = 0; q < kKOthX / 8; ++q) {
        *reinterpret_cast<uint4 *>(sXOh + trow * kLhXO + 8 * q) = z;
        *reinterpret_cast<uint4 *>(sXOl + trow * kLhXO + 8 * q) = z;
      }
    }
  }
  const int at0 = trow * kLhX2 + kLin;
  if (part == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) put_split(sX2h, sX2l, at0 + k, (float)tv[k]);
  } else if (part == 1) {
#pragma unroll
    for (int k = 0; k < kA; ++k)
      sX2h[at0 + 4 + k] = (_Float16)(float)(int)((k == 0 ? (prev_act_t == 0) | (prev_done_t != 0) : (prev_act_t == k) & (prev_done_t == 0)));
    sX2h[at0 + 9] = gl == 0 ? (_Float16)1.0f : (_Float16)0.0f;
    sX2h[at0 + 10] = gl == 0 ? (_Float16)0.0f : (_Float16)1.0f;
  } else if (part == 3) {
#pragma unroll
    for (int k = 0; k < kLoMax; ++k)
      if (k < p.Lo) put_split(sXOh, sXOl, trow * kLhXO + k, (float)ov2[k >> 1][k & 1]);
  }
  if (dwords) {
    // window bytes -> float16 (values in {-1, 0, 1}): each dword's four bytes go to their (row, k) slots
#pragma unroll
    for (int q = 0; q < kWin; ++q) {
      const int dd = wli + lpe * q;
#pragma unroll
      for (int sb = 0; sb < 4; ++sb) {
        const int bb = 4 * dd + sb;
        if (bb < rec) {
          const int i = bb / kObs, k = bb - i * kObs;
          sX0[(wel * N + i) * kLhX0 + k] = (_Float16)(float)(int8_t)(wv[q] >> (8 * sb));
        }
      }
    }
  } else {
    for (int idx = tid; idx < 64 * kKConvX; idx += 256) {
      const int r = idx / kKConvX, k = idx - r * kKConvX;
      size_t row = row_base + r;
      row = row < rows ? row : rows - 1;
      const size_t e = row / N;
      const int i = (int)(row - e * N);
      sX0[r * kLhX0 + k] = k < kObs ? (_Float16)(float)p.obs_self_t[e * (size_t)p.obst_stride + (size_t)i * kObs + k] : (_Float16)0.0f;
    }
  }
}

#ifndef CM3_NO_ENTRY_POINTS
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) k_ck_actor_x3(const CkActorParams p) {
  using namespace ck_actor;
  __shared__ __attribute__((aligned(16))) _Float16 sH[kCkX3HBytes / 2];
  __shared__ __attribute__((aligned(16))) _Float16 sX0[kCkX3X0Bytes / 2];
  __shared__ __attribute__((aligned(16))) _Float16 sX2[kCkX3X2Bytes / 2];
  __shared__ __attribute__((aligned(16))) _Float16 sXO[kCkX3XOBytes / 2];
  __shared__ float sLG[64][8];
  CkX3Planes L;
  L.Hh = sH; L.Hl = sH + 64 * kLdHb;
  L.X0 = sX0; L.C1h = sH; L.C1l = sH + 64 * kLhC1;
  L.X2h = sX2; L.X2l = sX2 + 64 * kLhX2; L.XOh = sXO; L.XOl = sXO + 64 * kLhXO;
  L.LG = sLG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t rows = (size_t)p.E * p.N;
  const size_t row_base = (size_t)blockIdx.x * 64;
  const float *pk = p.packed;
  CM3_STAMP(0, false);
  CkConvB b_conv;
  ck_x3_load_conv(pk, w, lane, b_conv);
  if (w < 4) ck_x3_stage_inputs(p, L, tid, row_base, rows);
  CM3_STAMP(1, true);
  __syncthreads();
  CM3_STAMP(2, false);
  f32x4 acc2[4][kCkBCT];
  ck_x3_h2_bias(pk, w, lane, acc2);
  if (p.stage > 1) ck_x3_others(L, pk, w, lane, acc2);
  ck_x3_self_chain(L, pk, w, lane, b_conv, acc2);
  if (w < 4) {
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    ck_actor_head(p, sLG, w, lane, row_base, rows);
  }
  CM3_STAMP(12, true);
}

// ---- the others branch as a table (round 6) -----------------------------------------------------------------------------------------
// With two agents a row's v_obs_others is the normalised (row, column) of the ONE other agent: (r - 3.5) / 7 and (c - 6.5) / 13 with
// r in 0..6, c in 0..12 (checkers.py:112-125, :139) -- 91 possible inputs, so branch_others W_others_h2 has 91 possible values per
// unit.  This kernel evaluates ck_x3_others() -- the very function of the forward pass -- on those 91 inputs (as 2 x 64 rows) and
// stores the accumulators: tab[r * 13 + c][unit], float32.  The whole-episode kernel starts h2's accumulators from that row instead
// of running the branch: the same bits, 11 k of a tick's ~45 k cycles less.
constexpr int kCkOthCells = 7 * 13;
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) k_ck_actor_others_table(const float *pk, float *tab) {
  using namespace ck_actor;
  __shared__ __attribute__((aligned(16))) _Float16 sH[kCkX3HBytes / 2];
  __shared__ __attribute__((aligned(16))) _Float16 sXO[kCkX3XOBytes / 2];
  CkX3Planes L;
  memset(&L, 0, sizeof(L));
  L.Hh = sH; L.Hl = sH + 64 * kLdHb;
  L.XOh = sXO; L.XOl = sXO + 64 * kLhXO;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int idx = tid; idx < 2 * 64 * kLhXO; idx += 512) sXO[idx] = (_Float16)0.0f;
  __syncthreads();
  if (tid < 64) {
    int cell = (int)blockIdx.x * 64 + tid;
    cell = cell < kCkOthCells ? cell : kCkOthCells - 1;
    const int r = cell / 13, c = cell - 13 * r;
    // the env's own expressions (csrc/checkers.hip CkBoardTab::norm), cast to float32 like a TF feed, split like ck_x3_stage_inputs
    const double vr = ((double)r - 7.0 / 2.0) / 7.0, vc = ((double)c - 13.0 / 2.0) / 13.0;
    put_split(L.XOh, L.XOl, tid * kLhXO + 0, (float)vr);
    put_split(L.XOh, L.XOl, tid * kLhXO + 1, (float)vc);
  }
  __syncthreads();
  f32x4 acc2[4][kCkBCT];
  ck_x3_h2_bias(pk, w, lane, acc2);
  ck_x3_others(L, pk, w, lane, acc2);
  const int col = lane & 15, hi = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int c = 0; c < kCkBCT; ++c) {
      const int cell = (int)blockIdx.x * 64 + 16 * t + col;
      if (cell < kCkOthCells) {
        float4 v;
        v.x = acc2[t][c][0]; v.y = acc2[t][c][1]; v.z = acc2[t][c][2]; v.w = acc2[t][c][3];
        *reinterpret_cast<float4 *>(tab + (size_t)cell * kH2 + 16 * (kCkBCT * w + c) + 4 * hi) = v;
      }
    }
}

#endif  // CM3_NO_ENTRY_POINTS

static int ck_actor_check(const cm3_actor_checkers_desc *d) {
  using namespace ck_actor;
  CM3_REQUIRE(d, "null desc");
  CM3_REQUIRE(d->n_agents >= 1 && d->n_agents <= 8, "Checkers actor: n_agents must be in 1..8");
  CM3_REQUIRE(d->conv_f == kConvF && d->n_conv_linear == kLin && d->n_h1 == kH1 && d->n_h2 == kH2 && d->n_actions == kA,
              "supported Checkers actor widths are conv_f 6 / conv_linear 32 / h1 256 / h2 256 / 5 actions "
              "(config_checkers_stage*.json nn block); got %d/%d/%d/%d/%d",
              d->conv_f, d->n_conv_linear, d->n_h1, d->n_h2, d->n_actions);
  CM3_REQUIRE(d->n_obs == 2, "the actor reads 5x5x3 windows (n_obs = 2); got n_obs = %d", d->n_obs);
  return CM3_OK;
}

static void ck_actor_weights(CkActorParams &p, const cm3_actor_checkers_weights *wt) {
  p.conv_w = wt->conv_w; p.conv_b = wt->conv_b; p.lin_w = wt->lin_w; p.lin_b = wt->lin_b;
  p.self_w = wt->self_w; p.self_b = wt->self_b; p.w_self_h2 = wt->w_self_h2;
  p.oth_w = wt->others_w; p.oth_b = wt->others_b; p.w_oth_h2 = wt->w_others_h2;
  p.b_h2 = wt->b_h2; p.out_w = wt->out_w; p.out_b = wt->out_b;
}

}  // namespace cm3

#ifndef CM3_NO_ENTRY_POINTS
extern "C" size_t cm3_actor_checkers_packed_bytes(void) { return (size_t)cm3::ck_actor::kPAll * sizeof(float); }

extern "C" int cm3_actor_checkers_pack(const cm3_actor_checkers_desc *d, const cm3_actor_checkers_weights *wt,
                                       void *packed, void *stream) {
  using namespace cm3;
  int rc = ck_actor_check(d);
  if (rc != CM3_OK) return rc;
  CM3_REQUIRE(wt && packed, "null weights / packed buffer");
  CM3_REQUIRE(wt->conv_w && wt->conv_b && wt->lin_w && wt->lin_b && wt->self_w && wt->self_b && wt->w_self_h2 &&
                  wt->b_h2 && wt->out_w && wt->out_b, "missing weights");
  if (d->stage > 1) CM3_REQUIRE(wt->others_w && wt->others_b && wt->w_others_h2, "stage 2 needs the others branch");
  CkActorParams p;
  memset(&p, 0, sizeof(p));
  p.N = d->n_agents;
  p.stage = d->stage;
  p.Lo = 2 * (d->n_agents > 1 ? d->n_agents - 1 : 1);
  ck_actor_weights(p, wt);
  hipLaunchKernelGGL(k_ck_actor_pack, dim3(128), dim3(256), 0, (hipStream_t)stream, p, (float *)packed);
  CM3_HIP_CHECK(hipGetLastError());
  if (d->stage > 1 && d->n_agents == 2) {   // the others branch as a table (see k_ck_actor_others_table), from the weights just packed
    hipLaunchKernelGGL(k_ck_actor_others_table, dim3(2), dim3(512), 0, (hipStream_t)stream, (const float *)packed,
                       (float *)packed + ck_actor::kPOthTab);
    CM3_HIP_CHECK(hipGetLastError());
  }
  return CM3_OK;
}

extern "C" int cm3_actor_checkers_f32(const cm3_actor_checkers_desc *d, const cm3_actor_checkers_weights *wt,
                                      const cm3_actor_checkers_bufs *b, void *stream) {
  using namespace cm3;
  int rc = ck_actor_check(d);
  if (rc != CM3_OK) return rc;
  CM3_REQUIRE(wt && b, "null weights/bufs");
  CM3_REQUIRE(d->n_envs > 0, "n_envs must be positive");
  CM3_REQUIRE(d->epsilon >= 0.0f && d->epsilon <= 1.0f, "epsilon must be in [0,1]");
  CM3_REQUIRE(d->precision >= 0 && d->precision <= 2, "precision must be 0 (float32), 1 (bf16 256x256 layers) or 2 (split float16)");
  CM3_REQUIRE(wt->packed, "weights->packed is NULL: run cm3_actor_checkers_pack once per weight update");
  CM3_REQUIRE(b->obs_self_t && b->obs_self_v && b->obs_others && b->goals && b->steps && b->episode && b->actions,
              "missing buffers");
  CM3_REQUIRE(d->obs_self_t_stride >= d->n_agents * ck_actor::kObs, "obs_self_t_stride %d smaller than one env record",
              d->obs_self_t_stride);
  CkActorParams p;
  memset(&p, 0, sizeof(p));
  p.E = d->n_envs;
  p.N = d->n_agents;
  p.stage = d->stage;
  p.Lo = 2 * (d->n_agents > 1 ? d->n_agents - 1 : 1);
  p.eps = d->epsilon;
  p.env_id_base = d->env_id_base;
  p.seed = d->seed;
  p.obst_stride = d->obs_self_t_stride;
  p.obs_self_t = b->obs_self_t;
  p.obs_self_v = b->obs_self_v;
  p.obs_others = b->obs_others;
  p.goals = b->goals;
  p.actions_prev = b->actions_prev;
  p.prev_done = b->prev_done;
  p.eps_dev = b->epsilon_dev;
  p.steps = b->steps;
  p.episode = b->episode;
  p.actions = b->actions;
  p.probs = b->probs;
  p.packed = (const float *)wt->packed;
  const size_t rows = (size_t)p.E * p.N;
  const dim3 grid((unsigned)((rows + 63) / 64));
  if (d->precision == 1) hipLaunchKernelGGL(k_ck_actor<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (d->precision == 2) hipLaunchKernelGGL(k_ck_actor_x3, grid, dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(k_ck_actor<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}
#endif  // CM3_NO_ENTRY_POINTS

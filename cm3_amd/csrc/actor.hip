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
#include "common.h"
#include "philox.h"

namespace cm3 {

constexpr int kH1S = 64, kH1O = 128, kH2 = 64, kA = 5;
constexpr uint32_t kPurposePolicy = 0x40000000u;

struct ActorParams {
  int E, stage;
  float eps;
  int bf16;  // second layer on the bf16 matrix cores (f32 accumulate); 0 = exact f32 MFMA
  int64_t env_id_base;
  uint64_t seed;
  const float *obs_others, *state, *goals;
  const int32_t *meta, *episode;
  int32_t *actions;
  float *probs;
  const float *w_self, *b_self, *w_self_h2, *w_oth, *b_oth, *w_oth_h2, *b_h2, *w_out, *b_out;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Final layer for one row per lane (wave 0 of the workgroup): actor_out + softmax (networks.py:536-537),
// epsilon mix (alg_credit.py:119) and inverse-CDF sampling (:120) from the build's Philox stream.
template <int N>
__device__ __forceinline__ void actor_head(const ActorParams &p, const float (*h2s)[kH2 + 1], int lane, size_t r,
                                           bool row_ok, size_t e, int i) {
  float o[kA];
#pragma unroll
  for (int a = 0; a < kA; ++a) o[a] = p.b_out[a];
#pragma unroll 4
  for (int k = 0; k < kH2; ++k) {
    const float hk = h2s[lane][k];
#pragma unroll
    for (int a = 0; a < kA; ++a) o[a] = fmaf(hk, p.w_out[k * kA + a], o[a]);
  }
  float m = o[0];
#pragma unroll
  for (int a = 1; a < kA; ++a) m = fmaxf(m, o[a]);
  float sum = 0.0f;
#pragma unroll
  for (int a = 0; a < kA; ++a) {
    o[a] = expf(o[a] - m);
    sum += o[a];
  }
  float pr[kA];
#pragma unroll
  for (int a = 0; a < kA; ++a) pr[a] = (1.0f - p.eps) * (o[a] / sum) + p.eps / (float)kA;

  const int steps = p.meta[2 * e];
  const uint32_t episode = (uint32_t)p.episode[e];
  const uint64_t genv = (uint64_t)(p.env_id_base + (int64_t)e);
  u32x4 ctr;
  ctr.x = (uint32_t)genv;
  ctr.y = (uint32_t)(genv >> 32);
  ctr.z = episode;
  ctr.w = kPurposePolicy | ((uint32_t)(i >> 2) << 24) | ((uint32_t)steps & 0x00FFFFFFu);
  const u32x4 wd = philox4x32_10(ctr, (uint32_t)p.seed, (uint32_t)(p.seed >> 32));
  const int q = i & 3;
  const float u = (float)u01(q == 0 ? wd.x : (q == 1 ? wd.y : (q == 2 ? wd.z : wd.w)));
  int act = kA - 1;
  float cdf = 0.0f;
  bool chosen = false;
#pragma unroll
  for (int a = 0; a < kA - 1; ++a) {
    cdf += pr[a];
    if (!chosen && u < cdf) {
      act = a;
      chosen = true;
    }
  }
  if (row_ok) {
    p.actions[r] = act;
    if (p.probs) {
#pragma unroll
      for (int a = 0; a < kA; ++a) p.probs[r * kA + a] = pr[a];
    }
  }
}

// Workgroup = 4 waves = 64 rows.
//   Phase A (VALU, one row per lane): wave w evaluates a quarter of the first-layer units -- 16 of branch_self and
//     32 of actor_others -- for all 64 rows; the unit weights are uniform across lanes and come from a transposed
//     LDS copy as broadcast reads.  Results go to LDS h1s[row][unit] (unit index = k of the second layer).
//   Phase B (matrix cores): wave w owns columns [16w, 16w+16) of the 192 -> 64 second layer,
//     v_mfma_f32_16x16x4_f32 (exact f32, k-ordered fma chain), 4 row tiles x 48 k-steps:
//       A[i = l&15][k = l>>4]  = h1s[16t + (l&15)][4s + (l>>4)]        one ds_read_b32 per MFMA
//       B[k = l>>4][j = l&15]  = W2[4s + (l>>4)][16w + (l&15)]         48 VGPRs per lane, loaded once
//       C  col = l&15, row = 4 (l>>4) + reg  -> relu(C + b) to LDS h2s, where wave 0 picks up whole rows.
//   (A version that evaluated the first layer directly in the A-operand layout needed no h1s but recomputed it in
//    all four waves: 2750 VALU instructions per wave, 14.5 us per workgroup -- PMC run in profiles/.)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// BF16 == true (opt-in, cm3_actor_particle_desc.precision = 1): the first-layer activations are stored as bf16 and
// the second layer runs on v_mfma_f32_16x16x32_bf16 (f32 accumulate): 6 k-steps instead of 48, 16x the MFMA rate.
//       A[i = l&15][k = 8 (l>>4) + q] = h1b[16t + (l&15)][32s + 8 (l>>4) + q],  q = 0..7   one ds_read_b128 per MFMA
//       B[k = 8 (l>>4) + q][j = l&15] = bf16(W2[32s + 8 (l>>4) + q][16w + (l&15)])          24 VGPRs per lane
// Inputs and W2 are rounded to bf16 (relative 2^-9): probabilities move by up to ~1e-2, so this is NOT the parity
// path; everything else (first layers, output layer, softmax, sampling) stays float32.
template <int N, bool BF16> __global__ void __launch_bounds__(256) k_actor_particle(const ActorParams p) {
  constexpr int L = 4 * (N > 1 ? N - 1 : 1);
  constexpr int SW = 8;                  // ws_self row: 6 weights, bias, pad
  constexpr int OW = L + 4;              // ws_oth row: L weights, bias, pad (multiple of 4 floats)
  constexpr int KU = kH1S + kH1O;        // 192 first-layer units = K of the second layer
  constexpr int HB = KU + 8;             // bf16 row: 200 halfwords = 400 B (16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) float ws_self[kH1S][SW];
  __shared__ __attribute__((aligned(16))) float ws_oth[kH1O][OW];
  __shared__ __attribute__((aligned(16))) float h1raw[BF16 ? (64 * HB) / 2 : 64 * (KU + 1)];
  float (*h1s)[KU + 1] = reinterpret_cast<float (*)[KU + 1]>(h1raw);       // f32 view  [64][193]
  __bf16 (*h1b)[HB] = reinterpret_cast<__bf16 (*)[HB]>(h1raw);             // bf16 view [64][200]
  __shared__ float h2s[64][kH2 + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, hi = lane >> 4;
  const int c0 = w * 16;
  const size_t rows = (size_t)p.E * N;
  const size_t row_base = (size_t)blockIdx.x * 64;
  const bool stage2 = p.stage > 1;

  // ---- first-layer weights -> LDS, transposed to [unit][input] --------------------------------------------------
  for (int idx = tid; idx < 6 * kH1S; idx += 256) ws_self[idx % kH1S][idx / kH1S] = p.w_self[idx];
  for (int k = tid; k < kH1S; k += 256) {
    ws_self[k][6] = p.b_self[k];
    ws_self[k][7] = 0.0f;
  }
  if (stage2) {
    for (int idx = tid; idx < L * kH1O; idx += 256) ws_oth[idx % kH1O][idx / kH1O] = p.w_oth[idx];
    for (int k = tid; k < kH1O; k += 256) {
      ws_oth[k][L] = p.b_oth[k];
      ws_oth[k][L + 1] = ws_oth[k][L + 2] = ws_oth[k][L + 3] = 0.0f;
    }
  }
  // ---- this lane's row: concat(v_obs, v_goal) and obs_others -----------------------------------------------------------
  const size_t r = row_base + lane;
  const bool row_ok = r < rows;
  const size_t rc = row_ok ? r : rows - 1;
  const size_t e = rc / N;
  const int i = (int)(rc - e * N);
  float x[6], xo[L];
  {
    const float4 s = reinterpret_cast<const float4 *>(p.state)[(size_t)i * p.E + e];
    const float2 g = reinterpret_cast<const float2 *>(p.goals)[(size_t)i * p.E + e];
    x[0] = s.x; x[1] = s.y; x[2] = s.z; x[3] = s.w; x[4] = g.x; x[5] = g.y;
    const float4 *o4 = reinterpret_cast<const float4 *>(p.obs_others + rc * L);
#pragma unroll
    for (int k = 0; k < L / 4; ++k) {
      const float4 v = o4[k];
      xo[4 * k + 0] = v.x; xo[4 * k + 1] = v.y; xo[4 * k + 2] = v.z; xo[4 * k + 3] = v.w;
    }
  }
  // ---- B: this wave's slice of W2 = [W_branch_self_h2 ; W_others_h2], unit k = 4s + hi, column c0 + col --------------
  float bw[BF16 ? 1 : KU / 4];
  bf16x8 bwb[BF16 ? KU / 32 : 1];
  if constexpr (BF16) {
#pragma unroll
    for (int s = 0; s < KU / 32; ++s)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = 32 * s + 8 * hi + q;  // unit index in [W_branch_self_h2 ; W_others_h2]
        const float wv = k < kH1S ? p.w_self_h2[k * kH2 + c0 + col]
                                  : (stage2 ? p.w_oth_h2[(k - kH1S) * kH2 + c0 + col] : 0.0f);
        bwb[s][q] = (__bf16)wv;
      }
  } else {
#pragma unroll
    for (int s = 0; s < kH1S / 4; ++s) bw[s] = p.w_self_h2[(4 * s + hi) * kH2 + c0 + col];
#pragma unroll
    for (int s = 0; s < kH1O / 4; ++s) bw[kH1S / 4 + s] = stage2 ? p.w_oth_h2[(4 * s + hi) * kH2 + c0 + col] : 0.0f;
  }
  __syncthreads();

  // ---- phase A: dense(6 -> 64, relu) units [16w, 16w+16) and dense(L -> 128, relu) units [32w, 32w+32) ---------------
#pragma unroll 4
  for (int q = 0; q < kH1S / 4; ++q) {  // networks.py:520-521
    const int k = 16 * w + q;
    const float4 wa = *reinterpret_cast<const float4 *>(&ws_self[k][0]);
    const float4 wb = *reinterpret_cast<const float4 *>(&ws_self[k][4]);
    float a = wb.z;  // bias
    a = fmaf(x[0], wa.x, a);
    a = fmaf(x[1], wa.y, a);
    a = fmaf(x[2], wa.z, a);
    a = fmaf(x[3], wa.w, a);
    a = fmaf(x[4], wb.x, a);
    a = fmaf(x[5], wb.y, a);
    if constexpr (BF16) h1b[lane][k] = (__bf16)fmaxf(a, 0.0f); else h1s[lane][k] = fmaxf(a, 0.0f);
  }
  if (stage2) {
#pragma unroll 4
    for (int q = 0; q < kH1O / 4; ++q) {  // networks.py:527-529
      const int k = 32 * w + q;
      float wv[OW];
#pragma unroll
      for (int c = 0; c < OW / 4; ++c) {
        const float4 v = *reinterpret_cast<const float4 *>(&ws_oth[k][4 * c]);
        wv[4 * c + 0] = v.x; wv[4 * c + 1] = v.y; wv[4 * c + 2] = v.z; wv[4 * c + 3] = v.w;
      }
      float a = wv[L];  // bias
#pragma unroll
      for (int c = 0; c < L; ++c) a = fmaf(xo[c], wv[c], a);
      if constexpr (BF16) h1b[lane][kH1S + k] = (__bf16)fmaxf(a, 0.0f); else h1s[lane][kH1S + k] = fmaxf(a, 0.0f);
    }
  } else {
#pragma unroll 4
    for (int q = 0; q < kH1O / 4; ++q) {
      if constexpr (BF16) h1b[lane][kH1S + 32 * w + q] = (__bf16)0.0f; else h1s[lane][kH1S + 32 * w + q] = 0.0f;
    }
  }
  __syncthreads();

  // ---- phase B: second layer on the matrix cores (networks.py:522-531: both matmuls, add_n) ---------------------------------
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (BF16) {
#pragma unroll
    for (int s = 0; s < KU / 32; ++s) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&h1b[16 * t + col][32 * s + 8 * hi]);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bwb[s], acc[t], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < KU / 4; ++s) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(h1s[16 * t + col][4 * s + hi], bw[s], acc[t], 0, 0, 0);
    }
  }
  // ---- h2 = relu(add_n + b) (networks.py:533-534): C tile -> LDS rows ---------------------------------------------------
  {
    const float bias = p.b_h2[c0 + col];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) h2s[16 * t + 4 * hi + reg][c0 + col] = fmaxf(acc[t][reg] + bias, 0.0f);
  }
  __syncthreads();
  if (w != 0) return;
  actor_head<N>(p, h2s, lane, r, row_ok, e, i);
}

template <int N> static int actor_launch(const ActorParams &p, hipStream_t s) {
  const size_t rows = (size_t)p.E * N;
  const unsigned blocks = (unsigned)((rows + 63) / 64);
  if (p.bf16)
    hipLaunchKernelGGL((k_actor_particle<N, true>), dim3(blocks), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((k_actor_particle<N, false>), dim3(blocks), dim3(256), 0, s, p);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

}  // namespace cm3

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
  CM3_REQUIRE(d->precision == 0 || d->precision == 1, "precision must be 0 (float32) or 1 (bf16 second layer)");
  CM3_REQUIRE(wt->w_self && wt->b_self && wt->w_self_h2 && wt->b_h2 && wt->w_out && wt->b_out, "missing weights");
  if (d->stage > 1) CM3_REQUIRE(wt->w_others && wt->b_others && wt->w_others_h2, "stage 2 needs the others branch");
  CM3_REQUIRE(b->obs_others && b->state && b->goals && b->meta && b->episode && b->actions, "missing buffers");
  ActorParams p;
  memset(&p, 0, sizeof(p));
  p.E = d->n_envs;
  p.stage = d->stage;
  p.eps = d->epsilon;
  p.bf16 = d->precision == 1 ? 1 : 0;
  p.env_id_base = d->env_id_base;
  p.seed = d->seed;
  p.obs_others = (const float *)b->obs_others;
  p.state = (const float *)b->state;
  p.goals = (const float *)b->goals;
  p.meta = b->meta;
  p.episode = b->episode;
  p.actions = b->actions;
  p.probs = b->probs;
  p.w_self = wt->w_self; p.b_self = wt->b_self; p.w_self_h2 = wt->w_self_h2;
  p.w_oth = wt->w_others; p.b_oth = wt->b_others; p.w_oth_h2 = wt->w_others_h2;
  p.b_h2 = wt->b_h2; p.w_out = wt->w_out; p.b_out = wt->b_out;
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
  }
  return fail(CM3_ERR_INVALID, "n_agents %d unsupported", d->n_agents);
}

// Data movement of the rows SURVEY.md section 8(f2) / (f4) as single launches (round 5):
//   cm3_transitions_gather_f32   every column of the reference's 11-field transition (alg/train_onpolicy.py:338, consumed by
//                                alg_credit.process_batch, alg_credit.py:458-470) for a list of (tick, env) pairs, straight out of the
//                                time-major trajectory: next_* = slot t + 1 or the captured terminal values where the env restarted
//                                in the same launch, goals from the slot that last wrote them (sparse goal slots) -- ONE launch
//                                instead of the ~25 indexing / where launches of the torch composition;
//   cm3_rows_scatter             the columns of a batch of transitions into ring positions (replay_buffer.py:11-16 `add`, and the
//                                dual buffer's two rings, replay_buffer_dual.py:12-38) -- all columns in one launch;
//   cm3_rows_gather              the columns of sampled transitions out of a ring (replay_buffer.py:28-37) -- all columns in one launch.
// Pure byte movement: HBM-bound, no arithmetic, no matrix cores.  A wave copies whole rows with 16-byte (or 8- / 4- / 1-byte, by the
// row size) units, consecutive lanes on consecutive units, so that the side with contiguous rows is fully coalesced.
#include "common.h"

namespace cm3 {

constexpr int kMaxRowCols = 16;

struct RowCols {
  int n;
  void *dst[kMaxRowCols];
  const void *src[kMaxRowCols];
  uint32_t row_bytes[kMaxRowCols];
  uint32_t unit[kMaxRowCols];   // bytes per lane: 16, 8, 4 or 1 (largest that divides the row size and both base addresses)
};

// (I = uint32_t whenever the unit count fits: a 64-bit division per unit costs more than the copy)
template <typename U, typename I> __device__ __forceinline__ void copy_units(void *dst, const void *src, size_t n_rows, uint32_t units_per_row,
                                                                             const int64_t *dst_row, const int64_t *src_row,
                                                                             int64_t ring_start, int64_t ring_size) {
  const I total = (I)(n_rows * units_per_row), stride = (I)((size_t)gridDim.x * blockDim.x);
  for (I g = (I)((size_t)blockIdx.x * blockDim.x + threadIdx.x); g < total; g += stride) {
    const I b = g / (I)units_per_row;
    const uint32_t u = (uint32_t)(g - b * (I)units_per_row);
    int64_t d = (int64_t)b, s = (int64_t)b;
    if (dst_row) {
      d = dst_row[b];
    } else if (ring_size > 0) {
      d = ring_start + (int64_t)b;
      d = d >= ring_size ? d - ring_size : d;
    }
    if (src_row) s = src_row[b];
    if (d < 0 || s < 0) continue;     // (a negative row index = skip: how the dual buffer's two rings share one flag pass)
    reinterpret_cast<U *>(dst)[(size_t)d * units_per_row + u] = reinterpret_cast<const U *>(src)[(size_t)s * units_per_row + u];
  }
}

template <typename I> __global__ void __launch_bounds__(256) k_rows_copy(const RowCols c, size_t n_rows, const int64_t *dst_row,
                                                                         const int64_t *src_row, int64_t ring_start, int64_t ring_size) {
  const int col = blockIdx.y;
  const uint32_t unit = c.unit[col], upr = c.row_bytes[col] / unit;
  if (unit == 16)
    copy_units<uint4, I>(c.dst[col], c.src[col], n_rows, upr, dst_row, src_row, ring_start, ring_size);
  else if (unit == 8)
    copy_units<uint2, I>(c.dst[col], c.src[col], n_rows, upr, dst_row, src_row, ring_start, ring_size);
  else if (unit == 4)
    copy_units<uint32_t, I>(c.dst[col], c.src[col], n_rows, upr, dst_row, src_row, ring_start, ring_size);
  else
    copy_units<uint8_t, I>(c.dst[col], c.src[col], n_rows, upr, dst_row, src_row, ring_start, ring_size);
}

static int rows_copy(const cm3_row_cols *cols, int64_t n_rows, const int64_t *dst_row, const int64_t *src_row, int64_t ring_start,
                     int64_t ring_size, hipStream_t s) {
  CM3_REQUIRE(cols && cols->n_cols >= 1 && cols->n_cols <= kMaxRowCols, "row columns: 1..%d columns", kMaxRowCols);
  CM3_REQUIRE(n_rows >= 0, "n_rows must be >= 0");
  if (n_rows == 0) return CM3_OK;
  RowCols c;
  memset(&c, 0, sizeof(c));
  c.n = cols->n_cols;
  size_t most = 0;
  for (int k = 0; k < c.n; ++k) {
    CM3_REQUIRE(cols->dst[k] && cols->src[k] && cols->row_bytes[k] > 0, "row columns: column %d is null / empty", k);
    c.dst[k] = cols->dst[k];
    c.src[k] = cols->src[k];
    c.row_bytes[k] = cols->row_bytes[k];
    const uintptr_t a = (uintptr_t)cols->dst[k] | (uintptr_t)cols->src[k] | (uintptr_t)cols->row_bytes[k];
    c.unit[k] = (a % 16 == 0) ? 16u : (a % 8 == 0) ? 8u : (a % 4 == 0) ? 4u : 1u;
    const size_t units = (size_t)n_rows * (c.row_bytes[k] / c.unit[k]);
    most = units > most ? units : most;
  }
  size_t blocks = (most + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  if (most + blocks * 256 < (size_t)1 << 32)
    hipLaunchKernelGGL(k_rows_copy<uint32_t>, dim3((unsigned)blocks, (unsigned)c.n), dim3(256), 0, s, c, (size_t)n_rows, dst_row, src_row,
                       ring_start, ring_size);
  else
    hipLaunchKernelGGL(k_rows_copy<size_t>, dim3((unsigned)blocks, (unsigned)c.n), dim3(256), 0, s, c, (size_t)n_rows, dst_row, src_row,
                       ring_start, ring_size);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

// ---- the reference's transition, gathered from the trajectory ---------------------------------------------------------------------
struct TransParams {
  const float *state, *obs, *reward_n, *reward, *term_state, *term_obs, *goals;
  const int32_t *actions, *goal_slot;
  const uint8_t *done;
  size_t st_state, st_obs, st_actions, st_reward_n, st_reward, st_done, st_term_state, st_term_obs, st_goals, st_goal_slot;   // bytes per tick
  const int64_t *tt, *ee;
  float *o_state, *o_obs, *o_reward_n, *o_reward, *o_next_state, *o_next_obs, *o_goals;
  int32_t *o_actions;
  uint8_t *o_done;
  size_t n, E;
  size_t ring_start, ring_size;   // ring_size > 0: transition b goes to output row (ring_start + b) mod ring_size (a replay ring), else to row b
  int N, L;
};

template <typename T> __device__ __forceinline__ const T *tick_of(const T *base, size_t stride, int64_t t) {
  return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + stride * (size_t)t);
}

// A lane per (transition, unit): unit u of transition b's record is
//   [0, N)                state row of agent u          16 B      state[t][u][e]            -> o_state[b][u]
//   [N, 2N)               next state                    16 B      done ? term_state[t] : state[t + 1]
//   [2N, 2N + NV)         obs_others, NV = N L / 4      16 B      obs[t][e][...]            -> o_obs[b][...]   (contiguous both sides)
//   [.., + NV)            next obs_others               16 B      done ? term_obs[t] : obs[t + 1]
//   then N goal pairs (8 B), N actions, N local rewards (4 B), the reward (4 B) and done (1 B)
// (round 5, first form: a wave per transition looping over its units -- 46 of 64 lanes busy at N = 4 and one dependent chain of
// index -> done -> data loads per wave at a time: 1.23 ms for the 1.35 M transitions of a C2 phase, hardly faster than the torch
// composition; flattened, every lane has its own chain in flight)
// (second form: each lane carries FOUR units at once -- their `done` bytes are requested together, then their data, then the stores:
// a unit is two dependent memory round trips, and with one unit per lane in flight the export ran at 2 TB/s)
struct TransUnit {
  const void *src;
  const void *alt;   // next_* units under terminal capture: the source when the env restarted at this tick (needs the done byte); else NULL
  void *dst;
  int bytes;   // 16, 8, 4, 1 (the done byte: written from the value already loaded), 0 (nothing: past the end)
};
__device__ __forceinline__ TransUnit trans_unit(const TransParams &p, size_t b_in, int v, int64_t t, size_t e, int N, int NV) {
  TransUnit u;
  u.alt = nullptr;
  size_t b = b_in;                // output row
  if (p.ring_size) {
    b = p.ring_start + b_in;
    b = b >= p.ring_size ? b - p.ring_size : b;
  }
  if (v < 2 * N) {
    const bool nxt = v >= N;
    const int i = nxt ? v - N : v;
    const float *src = nxt ? tick_of(p.state, p.st_state, t + 1) : tick_of(p.state, p.st_state, t);
    u.src = reinterpret_cast<const float4 *>(src) + ((size_t)i * p.E + e);
    if (nxt && p.term_state) u.alt = reinterpret_cast<const float4 *>(tick_of(p.term_state, p.st_term_state, t)) + ((size_t)i * p.E + e);
    u.dst = reinterpret_cast<float4 *>(nxt ? p.o_next_state : p.o_state) + (b * N + i);
    u.bytes = 16;
    return u;
  }
  v -= 2 * N;
  if (v < 2 * NV) {
    const bool nxt = v >= NV;
    const int k = nxt ? v - NV : v;
    const float *src = nxt ? tick_of(p.obs, p.st_obs, t + 1) : tick_of(p.obs, p.st_obs, t);
    u.src = reinterpret_cast<const float4 *>(src) + (e * NV + k);
    if (nxt && p.term_obs) u.alt = reinterpret_cast<const float4 *>(tick_of(p.term_obs, p.st_term_obs, t)) + (e * NV + k);
    u.dst = reinterpret_cast<float4 *>(nxt ? p.o_next_obs : p.o_obs) + (b * NV + k);
    u.bytes = 16;
    return u;
  }
  v -= 2 * NV;
  if (v < N) {   // goals [slot][N][E][2]: the slot that last wrote this env's landmarks (goal_slot), or the only one (stride 0)
    const int64_t gs = p.goal_slot ? (int64_t)tick_of(p.goal_slot, p.st_goal_slot, t)[e] : t;
    u.src = reinterpret_cast<const float2 *>(tick_of(p.goals, p.st_goals, gs)) + ((size_t)v * p.E + e);
    u.dst = reinterpret_cast<float2 *>(p.o_goals) + (b * N + v);
    u.bytes = 8;
    return u;
  }
  v -= N;
  if (v < N) {
    u.src = tick_of(p.actions, p.st_actions, t) + (e * N + v);
    u.dst = p.o_actions + (b * N + v);
    u.bytes = 4;
    return u;
  }
  v -= N;
  if (v < N) {
    u.src = tick_of(p.reward_n, p.st_reward_n, t) + (e * N + v);
    u.dst = p.o_reward_n + (b * N + v);
    u.bytes = 4;
    return u;
  }
  v -= N;
  if (v == 0) {
    u.src = tick_of(p.reward, p.st_reward, t) + e;
    u.dst = p.o_reward + b;
    u.bytes = 4;
    return u;
  }
  u.src = nullptr;
  u.dst = p.o_done + b;
  u.bytes = 1;
  return u;
}

__global__ void __launch_bounds__(256) k_transitions_gather(const TransParams p) {
  constexpr int K = 4;
  const int N = p.N, NV = N * p.L / 4;
  const uint32_t U = (uint32_t)(2 * N + 2 * NV + 3 * N + 2);
  const size_t total = p.n * U, stride = (size_t)gridDim.x * blockDim.x;
  const bool small = total + (size_t)K * stride < ((size_t)1 << 32);
  for (size_t g0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g0 < total; g0 += (size_t)K * stride) {
    size_t bb[K], ee[K];
    int64_t tt[K];
    int vv[K];
    uint8_t dd[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const size_t g = g0 + (size_t)k * stride, gc = g < total ? g : total - 1;
      size_t b;
      int v;
      if (p.tt) {   // an indexed gather: a transition's units on consecutive lanes
        b = small ? (size_t)((uint32_t)gc / U) : gc / U;
        v = (int)(gc - b * U);
      } else {
        // the whole trajectory in order (b = t E + e): the units whose SOURCE rows lie along the env index -- state, next state and
        // goals are [slot][N][E][..] -- take consecutive lanes on consecutive ENVS (whole lines read; the 16- / 8-byte pieces a
        // lane writes meet their neighbours of the same output line in the L2), the others stay transition-major.  With every
        // lane of a transition reading its own state row the export read 4 lines for 64 bytes of them and ran at 2.4 TB/s.
        const size_t nA = p.n * (size_t)(3 * N);
        if (gc < nA) {
          const size_t c = small ? (size_t)((uint32_t)gc / (uint32_t)p.n) : gc / p.n;
          b = gc - c * p.n;
          v = c < (size_t)(2 * N) ? (int)c : (int)c + 2 * NV;           // goals follow the observation units in the unit order
        } else {
          const size_t g2 = gc - nA;
          const uint32_t UB = (uint32_t)(2 * NV + 2 * N + 2);
          b = small ? (size_t)((uint32_t)g2 / UB) : g2 / UB;
          const int w2 = (int)(g2 - b * UB);
          v = w2 < 2 * NV ? 2 * N + w2 : w2 + 3 * N;                     // (2N + 2NV + N) + (w2 - 2NV)
        }
      }
      bb[k] = b;
      vv[k] = g < total ? v : -1;
      // (tt == NULL: every transition of the trajectory in time-major order, b = t E + e -- no index arrays to build or read)
      tt[k] = p.tt ? p.tt[b] : (int64_t)(small ? (size_t)((uint32_t)b / (uint32_t)p.E) : b / p.E);
      ee[k] = p.tt ? (size_t)p.ee[b] : b - (size_t)tt[k] * p.E;
    }
    TransUnit u[K];
    uint4 val[K];
    // Only the next_* units of a trajectory with terminal capture (and the done column itself) need the env's done byte before
    // their data can be requested: the byte is requested for those alone, every other unit's data in the same round trip
#pragma unroll
    for (int k = 0; k < K; ++k) {
      u[k] = trans_unit(p, bb[k], vv[k] < 0 ? 0 : vv[k], tt[k], ee[k], N, NV);
      if (vv[k] < 0) u[k].bytes = 0;
      dd[k] = 0;
      if (u[k].alt || u[k].bytes == 1) dd[k] = tick_of(p.done, p.st_done, tt[k])[ee[k]];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      val[k] = uint4{0u, 0u, 0u, 0u};
      if (!u[k].alt) {
        if (u[k].bytes == 16) val[k] = *reinterpret_cast<const uint4 *>(u[k].src);
        else if (u[k].bytes == 8) { const uint2 w = *reinterpret_cast<const uint2 *>(u[k].src); val[k].x = w.x; val[k].y = w.y; }
        else if (u[k].bytes == 4) val[k].x = *reinterpret_cast<const uint32_t *>(u[k].src);
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (u[k].alt) val[k] = *reinterpret_cast<const uint4 *>(dd[k] ? u[k].alt : u[k].src);      // (next state / next observation rows: 16 bytes)
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (u[k].bytes == 16) *reinterpret_cast<uint4 *>(u[k].dst) = val[k];
      else if (u[k].bytes == 8) *reinterpret_cast<uint2 *>(u[k].dst) = uint2{val[k].x, val[k].y};
      else if (u[k].bytes == 4) *reinterpret_cast<uint32_t *>(u[k].dst) = val[k].x;
      else if (u[k].bytes == 1) *reinterpret_cast<uint8_t *>(u[k].dst) = dd[k] ? 1 : 0;
    }
  }
}

// ---- tiling: the n x n credit repeats and the n x n x l_action counterfactual tiling of train_step ------------------------------
// (alg_credit.py:614-658 `np.repeat(np.reshape(x, [n_steps, n, d]), n, axis=0)` / `np.repeat(x, n, axis=0)`; :730-751 the same
// repeated l_action times against np.tile(np.eye(l_action)); process_actions / process_global_state :406-443, :528-557 with their
// `np.arange(N) != n` selections.)  Every output is "destination row r <- source row f(r)", f built from divisions and remainders of
// r; up to 16 outputs per launch.
struct TileCol {
  void *dst;
  const void *src;
  unsigned long long n_rows;      // destination rows
  uint32_t epr;                   // elements per row
  uint32_t kind;                  // CM3_TILE_*
  uint32_t es;                    // bytes per SOURCE element for CM3_TILE_COPY (1, 4, 8)
  uint32_t div0, mod0, mul0, div1, mod1, mul1;   // source row = ((r / div0) % mod0) * mul0 + ((r / div1) % mod1) * mul1   (mod 0: no remainder)
  uint32_t oth_n, oth_divq, oth_divn;            // oth_n = N > 0: source row = (r / divq) * N + j, j = k + (k >= n), k = r % (N - 1), n = (r / divn) % N
};
struct TileCols {
  int n;
  TileCol c[kMaxRowCols];
};

template <typename I> __global__ void __launch_bounds__(256) k_rows_tile(const TileCols a) {
  const TileCol &c = a.c[blockIdx.y];
  const I total = (I)((size_t)c.n_rows * c.epr), stride = (I)((size_t)gridDim.x * blockDim.x);
  for (I g = (I)((size_t)blockIdx.x * blockDim.x + threadIdx.x); g < total; g += stride) {
    const I r = g / (I)c.epr;
    const uint32_t e = (uint32_t)(g - r * (I)c.epr);
    I s;
    if (c.oth_n) {
      const uint32_t k = (uint32_t)(r % (I)(c.oth_n - 1)), n = (uint32_t)((r / (I)c.oth_divn) % (I)c.oth_n);
      s = (r / (I)c.oth_divq) * (I)c.oth_n + k + (k >= n ? 1u : 0u);
    } else {
      I q0 = r / (I)c.div0, q1 = r / (I)c.div1;
      if (c.mod0) q0 %= c.mod0;
      if (c.mod1) q1 %= c.mod1;
      s = q0 * c.mul0 + q1 * c.mul1;
    }
    switch (c.kind) {
      case CM3_TILE_COPY:
        if (c.es == 4)
          reinterpret_cast<uint32_t *>(c.dst)[g] = reinterpret_cast<const uint32_t *>(c.src)[s * c.epr + e];
        else if (c.es == 8)
          reinterpret_cast<uint64_t *>(c.dst)[g] = reinterpret_cast<const uint64_t *>(c.src)[s * c.epr + e];
        else
          reinterpret_cast<uint8_t *>(c.dst)[g] = reinterpret_cast<const uint8_t *>(c.src)[s * c.epr + e];
        break;
      case CM3_TILE_F32_TO_F64:
        reinterpret_cast<double *>(c.dst)[g] = (double)reinterpret_cast<const float *>(c.src)[s * c.epr + e];
        break;
      case CM3_TILE_ONEHOT_I64:
        reinterpret_cast<int64_t *>(c.dst)[g] = reinterpret_cast<const int32_t *>(c.src)[s] == (int32_t)e ? 1 : 0;
        break;
      case CM3_TILE_ONEHOT_F64:
        reinterpret_cast<double *>(c.dst)[g] = reinterpret_cast<const int32_t *>(c.src)[s] == (int32_t)e ? 1.0 : 0.0;
        break;
      case CM3_TILE_EYE_F64:
        reinterpret_cast<double *>(c.dst)[g] = (uint32_t)(r % c.epr) == e ? 1.0 : 0.0;
        break;
      default:   // CM3_TILE_NOT_I64: 1 - (byte != 0)   ("if true, then 0, else 1", alg_credit.py:590)
        reinterpret_cast<int64_t *>(c.dst)[g] = reinterpret_cast<const uint8_t *>(c.src)[s * c.epr + e] ? 0 : 1;
        break;
    }
  }
}

// TD target of the reference's train_step (alg_credit.py:594, :640, :684): reward + gamma * Q_target * done_multiplier, evaluated in
// NumPy's order -- (gamma * q) first, times the 0 / 1 multiplier, plus the float64 reward -- with no contraction (the library is built
// with -ffp-contract=off): the same bits as the torch composition in cm3_amd/batch.py, ONE launch instead of four
template <typename R> __global__ void __launch_bounds__(256) k_td_target(const R *reward, const double *q, const int64_t *mult, double gamma,
                                                                         double *out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const double gq = gamma * q[i];
    out[i] = (double)reward[i] + gq * (double)mult[i];
  }
}

}  // namespace cm3

extern "C" {
int cm3_td_target_f64(const void *reward, int32_t reward_is_f64, const double *q, const int64_t *multiplier, double gamma, double *out,
                      int64_t n, void *stream) {
  using namespace cm3;
  CM3_REQUIRE(n >= 0, "n must be >= 0");
  if (n == 0) return CM3_OK;
  CM3_REQUIRE(reward && q && multiplier && out, "td_target: null argument");
  size_t blocks = ((size_t)n + 255) / 256;
  blocks = blocks > 2048 ? 2048 : blocks;
  if (reward_is_f64)
    hipLaunchKernelGGL(k_td_target<double>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const double *)reward, q, multiplier,
                       gamma, out, (size_t)n);
  else
    hipLaunchKernelGGL(k_td_target<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float *)reward, q, multiplier,
                       gamma, out, (size_t)n);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

int cm3_rows_tile(const cm3_tile_col *cols, int32_t n_cols, void *stream) {
  using namespace cm3;
  CM3_REQUIRE(cols && n_cols >= 1 && n_cols <= kMaxRowCols, "rows_tile: 1..%d columns", kMaxRowCols);
  TileCols a;
  memset(&a, 0, sizeof(a));
  a.n = n_cols;
  size_t most = 0;
  for (int k = 0; k < n_cols; ++k) {
    const cm3_tile_col &q = cols[k];
    // (an EMPTY column may come with null pointers: torch.empty(0).data_ptr() is 0 -- ADVICE r5)
    CM3_REQUIRE(q.n_rows == 0 || (q.dst && (q.src || q.kind == CM3_TILE_EYE_F64)), "rows_tile: column %d is null", k);
    CM3_REQUIRE(q.kind >= CM3_TILE_COPY && q.kind <= CM3_TILE_NOT_I64, "rows_tile: column %d: unknown kind %d", k, (int)q.kind);
    CM3_REQUIRE(q.elems_per_row >= 1 && q.n_rows >= 0, "rows_tile: column %d: empty rows", k);
    CM3_REQUIRE(q.kind != CM3_TILE_COPY || q.elem_bytes == 1 || q.elem_bytes == 4 || q.elem_bytes == 8,
                "rows_tile: column %d: elem_bytes must be 1, 4 or 8", k);
    CM3_REQUIRE(q.others_n == 0 || (q.others_n >= 2 && q.others_divq >= 1 && q.others_divn >= 1), "rows_tile: column %d: bad others spec", k);
    CM3_REQUIRE(q.others_n != 0 || (q.div[0] >= 1 && q.div[1] >= 1), "rows_tile: column %d: divisors must be >= 1", k);
    TileCol &c = a.c[k];
    c.dst = q.dst;
    c.src = q.src;
    c.n_rows = (unsigned long long)q.n_rows;
    c.epr = q.elems_per_row;
    c.kind = q.kind;
    c.es = q.elem_bytes;
    c.div0 = q.div[0]; c.mod0 = q.mod[0]; c.mul0 = q.mul[0];
    c.div1 = q.div[1]; c.mod1 = q.mod[1]; c.mul1 = q.mul[1];
    c.oth_n = q.others_n; c.oth_divq = q.others_divq; c.oth_divn = q.others_divn;
    const size_t total = (size_t)q.n_rows * q.elems_per_row;
    most = total > most ? total : most;
  }
  if (most == 0) return CM3_OK;
  size_t blocks = (most + 255) / 256;
  blocks = blocks > 2048 ? 2048 : blocks;
  if (most + blocks * 256 < (size_t)1 << 32)
    hipLaunchKernelGGL(k_rows_tile<uint32_t>, dim3((unsigned)blocks, (unsigned)n_cols), dim3(256), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(k_rows_tile<size_t>, dim3((unsigned)blocks, (unsigned)n_cols), dim3(256), 0, (hipStream_t)stream, a);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

int cm3_rows_scatter(const cm3_row_cols *cols, int64_t n_rows, const int64_t *dst_row, int64_t ring_start, int64_t ring_size,
                     void *stream) {
  using namespace cm3;
  CM3_REQUIRE(dst_row || (ring_size > 0 && ring_start >= 0 && ring_start < ring_size && n_rows <= ring_size),
              "rows_scatter: either dst_row or a ring (0 <= ring_start < ring_size, n_rows <= ring_size)");
  return rows_copy(cols, n_rows, dst_row, nullptr, ring_start, dst_row ? 0 : ring_size, (hipStream_t)stream);
}

int cm3_rows_gather(const cm3_row_cols *cols, int64_t n_rows, const int64_t *src_row, void *stream) {
  using namespace cm3;
  CM3_REQUIRE(src_row, "rows_gather: src_row is NULL");
  return rows_copy(cols, n_rows, nullptr, src_row, 0, 0, (hipStream_t)stream);
}

int cm3_transitions_gather_f32(const cm3_particle_desc *desc, const cm3_particle_traj *traj, const int32_t *goal_slot,
                               size_t goal_slot_stride, const int64_t *tt, const int64_t *ee, int64_t n,
                               const cm3_transition_cols *out, void *stream) {
  using namespace cm3;
  CM3_REQUIRE(desc && traj && out && ((tt == nullptr) == (ee == nullptr)), "null argument (tt and ee: both or neither)");
  CM3_REQUIRE(n >= 0, "n must be >= 0");
  if (n == 0) return CM3_OK;      // (before the pointer checks: the columns of an empty batch are null -- ADVICE r5)
  CM3_REQUIRE(desc->n_agents >= 1 && desc->n_agents <= CM3_MAX_AGENTS, "n_agents out of range");
  CM3_REQUIRE(traj->state && traj->obs_others && traj->actions && traj->reward_n && traj->reward && traj->done && traj->goals,
              "trajectory base pointers are required");
  CM3_REQUIRE((traj->term_state == nullptr) == (traj->term_obs_others == nullptr), "term_state and term_obs_others: both or neither");
  CM3_REQUIRE(out->state && out->obs_others && out->actions && out->reward && out->reward_n && out->next_state && out->next_obs_others &&
                  out->done && out->goals,
              "output columns are required");
  TransParams p;
  memset(&p, 0, sizeof(p));
  p.state = (const float *)traj->state;             p.st_state = traj->state_stride;
  p.obs = (const float *)traj->obs_others;          p.st_obs = traj->obs_others_stride;
  p.actions = traj->actions;                        p.st_actions = traj->actions_stride;
  p.reward_n = (const float *)traj->reward_n;       p.st_reward_n = traj->reward_n_stride;
  p.reward = (const float *)traj->reward;           p.st_reward = traj->reward_stride;
  p.done = traj->done;                              p.st_done = traj->done_stride;
  p.term_state = (const float *)traj->term_state;   p.st_term_state = traj->term_state_stride;
  p.term_obs = (const float *)traj->term_obs_others; p.st_term_obs = traj->term_obs_others_stride;
  p.goals = (const float *)traj->goals;             p.st_goals = traj->goals_stride;
  p.goal_slot = goal_slot;                          p.st_goal_slot = goal_slot_stride;
  p.tt = tt;
  p.ee = ee;
  p.o_state = (float *)out->state;
  p.o_obs = (float *)out->obs_others;
  p.o_actions = out->actions;
  p.o_reward = (float *)out->reward;
  p.o_reward_n = (float *)out->reward_n;
  p.o_next_state = (float *)out->next_state;
  p.o_next_obs = (float *)out->next_obs_others;
  p.o_done = out->done;
  p.o_goals = (float *)out->goals;
  p.n = (size_t)n;
  CM3_REQUIRE(out->ring_size >= 0 && out->ring_start >= 0 && (out->ring_size == 0 || (out->ring_start < out->ring_size && n <= out->ring_size)),
              "transition columns: ring_start / ring_size out of range (0 <= ring_start < ring_size, n <= ring_size)");
  p.ring_start = (size_t)out->ring_start;
  p.ring_size = (size_t)out->ring_size;
  p.E = (size_t)desc->n_envs;
  p.N = desc->n_agents;
  p.L = 4 * (desc->n_agents > 1 ? desc->n_agents - 1 : 1);
  const size_t units = (size_t)n * (size_t)(2 * p.N + 2 * (p.N * p.L / 4) + 3 * p.N + 2);
  size_t blocks = (units + 4 * 256 - 1) / (4 * 256);     // four units per lane
  blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
  hipLaunchKernelGGL(k_transitions_gather, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}
}

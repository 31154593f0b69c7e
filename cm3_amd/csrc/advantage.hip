// Discounted returns over a device trajectory + the batch moments behind advantage normalisation.
//
// Build-defined extension: the reference's advantage (alg_credit.py:334-357) is not
// normalised and has no batch-wide statistic; north_star asks for an RCCL all-gather "only for the
// advantage-normalisation step that needs batch-wide stats".  Per rank this file produces
// (sum, sum of squares, count) in float64 DETERMINISTICALLY (fixed-shape two-level reduction, no atomics), the
// host all-gathers the three numbers (cm3_amd/shard.py) and cm3_normalize_* applies the global statistics.
#include "common.h"

namespace cm3 {

constexpr int kAdvBlock = 256;
constexpr int kAdvMaxBlocks = 1024;

// One thread per (env, channel) column, walking time backwards:
//   G[t] = x[t] + gamma * (1 - done[t]) * G[t+1],   G[T] = 0.
// The column is read in chunks of kAdvChunk time steps: all loads of a chunk are in flight before the (serial) recurrence
// consumes them -- 33 ticks cost 5 memory trips instead of 33 (15.5 + 4.4 us for two launches -> 10 us at 4096 envs x 4 agents).
// The block that finishes last (ticket from one atomic counter at the end of `scratch`, which it resets) folds the
// per-block partials in block order, so the result does not depend on which block that is: deterministic, one launch.
constexpr int kAdvChunk = 8;

template <typename R>
__global__ void __launch_bounds__(kAdvBlock) k_returns_moments(const R *__restrict__ x, const uint8_t *__restrict__ done,
                                                               const uint8_t *__restrict__ valid, R *__restrict__ out,
                                                               double *__restrict__ partials, double *__restrict__ moments,
                                                               int T, int E, int C, R gamma) {
  const size_t cols = (size_t)E * C;
  double s = 0.0, s2 = 0.0, n = 0.0;
  for (size_t col = (size_t)blockIdx.x * kAdvBlock + threadIdx.x; col < cols; col += (size_t)gridDim.x * kAdvBlock) {
    const size_t e = col / C;
    R g = R(0);
    for (int t_hi = T; t_hi > 0; t_hi -= kAdvChunk) {
      R xs[kAdvChunk];
      uint8_t ds[kAdvChunk], vs[kAdvChunk];
#pragma unroll
      for (int k = 0; k < kAdvChunk; ++k) {
        const int t = t_hi - 1 - k;
        const int tc = t >= 0 ? t : 0;
        xs[k] = x[(size_t)tc * cols + col];
        ds[k] = done[(size_t)tc * E + e];
        vs[k] = valid ? valid[(size_t)tc * E + e] : (uint8_t)1;
      }
#pragma unroll
      for (int k = 0; k < kAdvChunk; ++k) {
        const int t = t_hi - 1 - k;
        if (t >= 0) {
          g = xs[k] + (ds[k] != 0 ? R(0) : gamma * g);
          const bool v = vs[k] != 0;
          out[(size_t)t * cols + col] = v ? g : R(0);
          if (v) {
            const double gd = (double)g;
            s += gd;
            s2 += gd * gd;
            n += 1.0;
          }
        }
      }
    }
  }
  // fixed-shape reduction: wave shuffles, then the block's waves in order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    s2 += __shfl_down(s2, off, 64);
    n += __shfl_down(n, off, 64);
  }
  __shared__ double part[kAdvBlock / 64][3];
  __shared__ unsigned ticket;
  if ((threadIdx.x & 63) == 0) {
    part[threadIdx.x >> 6][0] = s;
    part[threadIdx.x >> 6][1] = s2;
    part[threadIdx.x >> 6][2] = n;
  }
  __syncthreads();
  unsigned *counter = reinterpret_cast<unsigned *>(partials + 3 * kAdvMaxBlocks);
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int w = 0; w < kAdvBlock / 64; ++w) {
      a += part[w][0];
      b += part[w][1];
      c += part[w][2];
    }
    partials[3 * blockIdx.x + 0] = a;
    partials[3 * blockIdx.x + 1] = b;
    partials[3 * blockIdx.x + 2] = c;
    __threadfence();
    ticket = atomicAdd(counter, 1u);
  }
  __syncthreads();
  if (ticket != gridDim.x - 1) return;
  // last block: every other block's partials are visible (fence before its ticket); wave 0 folds them in block order
  if (threadIdx.x < 64) {
    __threadfence();
    double fs = 0.0, fs2 = 0.0, fn = 0.0;
    const volatile double *pv = partials;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += 64) {
      fs += pv[3 * b + 0];
      fs2 += pv[3 * b + 1];
      fn += pv[3 * b + 2];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      fs += __shfl_down(fs, off, 64);
      fs2 += __shfl_down(fs2, off, 64);
      fn += __shfl_down(fn, off, 64);
    }
    if (threadIdx.x == 0) {
      moments[0] = fs;
      moments[1] = fs2;
      moments[2] = fn;
      *counter = 0u;  // every block has arrived: the counter is ready for the next call (no memset launch per call)
    }
  }
}

// x <- (x - mean) / (std + eps) with the statistics of ALL ranks: `parts` holds n_parts (sum, sum of squares, count)
// triples (this rank's, or the all-gathered ones), summed here in rank order -- every rank computes identical values.
// stats (optional) receives (mean, std, count).
// Segments (blockIdx.y; round 4): x is n_seg consecutive slabs of n_elem elements -- the rollouts of one collection phase -- each
// normalised with its OWN statistics; parts is [n_parts ranks][n_seg][3] (what one all-gather of every rank's [n_seg][3] delivers),
// stats [n_seg][3].  One segment = the original call.
template <typename R>
__global__ void __launch_bounds__(kAdvBlock) k_normalize(R *__restrict__ x, const uint8_t *__restrict__ valid,
                                                         const double *__restrict__ parts, int n_parts,
                                                         double *__restrict__ stats, size_t n_elem, int C, double eps,
                                                         int apply) {
  const int seg = blockIdx.y, n_seg = gridDim.y;
  x += (size_t)seg * n_elem;
  if (valid) valid += (size_t)seg * (n_elem / C);
  if (stats) stats += 3 * seg;
  parts += 3 * seg;
  const int rs = 3 * n_seg;   // stride between two ranks' triples of this segment
  double tot0 = parts[0], tot1 = parts[1], tot2 = parts[2];
  for (int r = 1; r < n_parts; ++r) {
    tot0 = tot0 + parts[rs * r + 0];
    tot1 = tot1 + parts[rs * r + 1];
    tot2 = tot2 + parts[rs * r + 2];
  }
  const double cnt = tot2 > 1.0 ? tot2 : 1.0;
  const double mean = tot0 / cnt;
  double var = tot1 / cnt - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double sd = ::sqrt(var);
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) {
    stats[0] = mean;
    stats[1] = sd;
    stats[2] = tot2;
  }
  if (!apply) return;
  // the same expression as the host helper cm3_amd.shard.normalize_advantages: (x - (R)mean) / (R)(std + eps), one IEEE
  // division per element in the working precision -- the two documented paths agree bit for bit
  const R m = (R)mean, den = (R)(sd + eps);
  for (size_t i = (size_t)blockIdx.x * kAdvBlock + threadIdx.x; i < n_elem; i += (size_t)gridDim.x * kAdvBlock) {
    const bool v = valid ? (valid[i / C] != 0) : true;
    x[i] = v ? (x[i] - m) / den : R(0);
  }
}

// several small device-to-device copies in ONE launch (trajectory slot <-> live env buffers): block b copies a slice of
// every region
struct CopyList {
  int n;
  void *dst[8];
  const void *src[8];
  size_t bytes[8];  // multiples of 16
};
__global__ void __launch_bounds__(256) k_copy_list(const CopyList c) {
  for (int r = 0; r < c.n; ++r) {
    const size_t n16 = c.bytes[r] >> 4;
    const uint4 *s = reinterpret_cast<const uint4 *>(c.src[r]);
    uint4 *d = reinterpret_cast<uint4 *>(c.dst[r]);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d[i] = s[i];
  }
}

// ---- the advantage step of ONE rank (the all-gather is the identity) as two launches, with the slot bookkeeping of the rollout ----
// A C4 rollout is a chain of dependent launches of ~2.6 us each; its tail used to be four of them (slot copy out, returns + moments,
// normalise -- and the slot copy in at the head).  Here: launch A = (optional cm3_copy_shift) + returns -> `out` + per-block partial
// moments; launch B = every block folds the partials in block order (exactly k_returns_moments' last-block fold, so the moments are
// the same bits), derives mean / std as k_normalize does and normalises its slice.  No atomics, no fences, no arrival counter.
// (Tried first and dropped: ONE launch with a grid-wide wait between the two halves.  Every block's release fence writes its XCD's
//  L2 back, 32 blocks per XCD one after the other: 21-54 us per launch under the profiler, C4 3.38 -> 3.51-4.5 us per tick.)
struct CopyShift {
  int n;
  void *first_dst[4];
  void *mid[4];
  const void *last_src[4];
  size_t bytes[4];
};

__device__ __forceinline__ void copy_shift_run(const CopyShift &cs) {
  // first_dst <- mid, then mid <- last_src, element by element (a lane owns an element of all three)
  for (int r = 0; r < cs.n; ++r) {
    const size_t n16 = cs.bytes[r] >> 4;
    uint4 *a = reinterpret_cast<uint4 *>(cs.first_dst[r]);
    uint4 *m = reinterpret_cast<uint4 *>(cs.mid[r]);
    const uint4 *c = reinterpret_cast<const uint4 *>(cs.last_src[r]);
    for (size_t i = (size_t)blockIdx.x * kAdvBlock + threadIdx.x; i < n16; i += (size_t)gridDim.x * kAdvBlock) {
      const uint4 keep = m[i], next = c[i];
      a[i] = keep;
      m[i] = next;
    }
  }
}

__device__ __forceinline__ void block_partials(double s, double s2, double n, double *__restrict__ partials) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    s2 += __shfl_down(s2, off, 64);
    n += __shfl_down(n, off, 64);
  }
  __shared__ double part[kAdvBlock / 64][3];
  if ((threadIdx.x & 63) == 0) {
    part[threadIdx.x >> 6][0] = s;
    part[threadIdx.x >> 6][1] = s2;
    part[threadIdx.x >> 6][2] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int w = 0; w < kAdvBlock / 64; ++w) {
      a += part[w][0];
      b += part[w][1];
      c += part[w][2];
    }
    partials[3 * blockIdx.x + 0] = a;
    partials[3 * blockIdx.x + 1] = b;
    partials[3 * blockIdx.x + 2] = c;
  }
}

// launch A, any T: k_returns_moments' column walk without its ticket
template <typename R>
__global__ void __launch_bounds__(kAdvBlock) k_returns_partials(const R *__restrict__ x, const uint8_t *__restrict__ done,
                                                                const uint8_t *__restrict__ valid, R *__restrict__ out,
                                                                double *__restrict__ partials, int T, int E, int C, R gamma,
                                                                const CopyShift cs) {
  if (blockIdx.y == 0) copy_shift_run(cs);
  const size_t cols = (size_t)E * C;
  {  // segment blockIdx.y: T ticks of the time-major arrays starting at tick blockIdx.y * T; its partials follow the previous segment's
    const size_t seg = blockIdx.y;
    x += seg * T * cols;
    out += seg * T * cols;
    done += seg * T * (size_t)E;
    if (valid) valid += seg * T * (size_t)E;
    partials += seg * 3 * gridDim.x;
  }
  double s = 0.0, s2 = 0.0, n = 0.0;
  for (size_t col = (size_t)blockIdx.x * kAdvBlock + threadIdx.x; col < cols; col += (size_t)gridDim.x * kAdvBlock) {
    const size_t e = col / C;
    R g = R(0);
    for (int t_hi = T; t_hi > 0; t_hi -= kAdvChunk) {
      R xs[kAdvChunk];
      uint8_t ds[kAdvChunk], vs[kAdvChunk];
#pragma unroll
      for (int k = 0; k < kAdvChunk; ++k) {
        const int t = t_hi - 1 - k;
        const int tc = t >= 0 ? t : 0;
        xs[k] = x[(size_t)tc * cols + col];
        ds[k] = done[(size_t)tc * E + e];
        vs[k] = valid ? valid[(size_t)tc * E + e] : (uint8_t)1;
      }
#pragma unroll
      for (int k = 0; k < kAdvChunk; ++k) {
        const int t = t_hi - 1 - k;
        if (t >= 0) {
          g = xs[k] + (ds[k] != 0 ? R(0) : gamma * g);
          const bool v = vs[k] != 0;
          out[(size_t)t * cols + col] = v ? g : R(0);
          if (v) {
            const double gd = (double)g;
            s += gd;
            s2 += gd * gd;
            n += 1.0;
          }
        }
      }
    }
  }
  block_partials(s, s2, n, partials);
}

// launch A for short trajectories (T <= TK, one column per lane): the whole column is loaded in ONE memory round trip (the walk
// above takes ceil(T / 8) dependent ones: 5 at T = 33).  Blocks beyond the column blocks only help with the slot bookkeeping; their
// partial moments are zeros appended to the fold, which leaves every lane's sum -- and so the moments -- bit-unchanged.
constexpr int kKeepTicks = 40;
constexpr int kKeepCopyBlocks = 256;

template <typename R, int TK>
__global__ void __launch_bounds__(kAdvBlock) k_returns_partials_keep(const R *__restrict__ x, const uint8_t *__restrict__ done,
                                                                     const uint8_t *__restrict__ valid, R *__restrict__ out,
                                                                     double *__restrict__ partials, int T, int E, int C, R gamma,
                                                                     int col_blocks, const CopyShift cs) {
  static_assert(TK <= 64, "done / valid are kept as bit masks");
  const size_t cols = (size_t)E * C;
  {  // segment blockIdx.y (see k_returns_partials)
    const size_t seg = blockIdx.y;
    x += seg * T * cols;
    out += seg * T * cols;
    done += seg * T * (size_t)E;
    if (valid) valid += seg * T * (size_t)E;
    partials += seg * 3 * gridDim.x;
  }
  const size_t col = (size_t)blockIdx.x * kAdvBlock + threadIdx.x;
  const bool act = (int)blockIdx.x < col_blocks && col < cols;
  const size_t cc = act ? col : 0, e = cc / C;
  R xs[TK];
  unsigned long long dm = 0ull, vm = ~0ull;
#pragma unroll
  for (int t = 0; t < TK; ++t) xs[t] = x[(size_t)(t < T ? t : T - 1) * cols + cc];
  {
    uint8_t ds[TK];
#pragma unroll
    for (int t = 0; t < TK; ++t) ds[t] = done[(size_t)(t < T ? t : T - 1) * E + e];
    if (valid) {
      uint8_t vs[TK];
#pragma unroll
      for (int t = 0; t < TK; ++t) vs[t] = valid[(size_t)(t < T ? t : T - 1) * E + e];
      vm = 0ull;
#pragma unroll
      for (int t = 0; t < TK; ++t) vm |= (unsigned long long)(vs[t] != 0) << t;
    }
    if (blockIdx.y == 0) copy_shift_run(cs);   // while the column is in flight
#pragma unroll
    for (int t = 0; t < TK; ++t) dm |= (unsigned long long)(ds[t] != 0) << t;
  }
  double s = 0.0, s2 = 0.0, n = 0.0;
  R g = R(0);
#pragma unroll
  for (int t = TK - 1; t >= 0; --t) {
    if (t < T) {
      g = xs[t] + (((dm >> t) & 1ull) ? R(0) : gamma * g);
      const bool v = ((vm >> t) & 1ull) != 0;
      if (act) out[(size_t)t * cols + col] = v ? g : R(0);
      if (v && act) {
        const double gd = (double)g;
        s += gd;
        s2 += gd * gd;
        n += 1.0;
      }
    }
  }
  block_partials(s, s2, n, partials);
}

// launch B: fold (block order, the shape of k_returns_moments' last block), statistics (k_normalize's expressions), normalisation
template <typename R>
__global__ void __launch_bounds__(kAdvBlock) k_fold_normalize(R *__restrict__ x, const uint8_t *__restrict__ valid,
                                                              const double *__restrict__ partials, int n_partials,
                                                              double *__restrict__ moments, double *__restrict__ stats,
                                                              size_t n_elem, int C, double eps, int apply) {
  {  // segment blockIdx.y: its slab of n_elem returns, its n_partials partial triples, its moments / stats
    const size_t seg = blockIdx.y;
    x += seg * n_elem;
    if (valid) valid += seg * (n_elem / C);
    partials += seg * 3 * n_partials;
    moments += 3 * seg;
    if (stats) stats += 3 * seg;
  }
  __shared__ double tot[3];
  if (threadIdx.x < 64) {
    double fs = 0.0, fs2 = 0.0, fn = 0.0;
    for (int b = threadIdx.x; b < n_partials; b += 64) {
      fs += partials[3 * b + 0];
      fs2 += partials[3 * b + 1];
      fn += partials[3 * b + 2];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      fs += __shfl_down(fs, off, 64);
      fs2 += __shfl_down(fs2, off, 64);
      fn += __shfl_down(fn, off, 64);
    }
    if (threadIdx.x == 0) {
      tot[0] = fs;
      tot[1] = fs2;
      tot[2] = fn;
    }
  }
  __syncthreads();
  const double tot0 = tot[0], tot1 = tot[1], tot2 = tot[2];
  const double cnt = tot2 > 1.0 ? tot2 : 1.0;
  const double mean = tot0 / cnt;
  double var = tot1 / cnt - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const double sd = ::sqrt(var);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    moments[0] = tot0;
    moments[1] = tot1;
    moments[2] = tot2;
    if (stats) {
      stats[0] = mean;
      stats[1] = sd;
      stats[2] = tot2;
    }
  }
  if (!apply) return;
  const R m = (R)mean, den = (R)(sd + eps);
  if constexpr (sizeof(R) == 4) {
    // four elements per lane and trip where nothing is masked and the slab is 16-byte aligned (round 4: one element per lane in
    // 2048 blocks per segment spent its time on the per-block fold -- 26 us per 330-tick C4 phase for 43 MB -- not on the elements)
    if (!valid && (n_elem & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
      float4 *x4 = reinterpret_cast<float4 *>(x);
      const size_t n4 = n_elem >> 2;
      for (size_t i = (size_t)blockIdx.x * kAdvBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kAdvBlock) {
        float4 v = x4[i];
        v.x = (v.x - m) / den;
        v.y = (v.y - m) / den;
        v.z = (v.z - m) / den;
        v.w = (v.w - m) / den;
        x4[i] = v;
      }
      return;
    }
  }
  for (size_t i = (size_t)blockIdx.x * kAdvBlock + threadIdx.x; i < n_elem; i += (size_t)gridDim.x * kAdvBlock) {
    const bool v = valid ? (valid[i / C] != 0) : true;
    x[i] = v ? (x[i] - m) / den : R(0);
  }
}

template <typename R>
static int returns_normalize(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                             double *moments, double *stats, int T, int E, int C, double gamma, double eps, int apply,
                             const cm3_copy_shift *shift, void *stream, int n_seg = 1) {
  CM3_REQUIRE(x && done && out && scratch && moments, "null pointer");
  CM3_REQUIRE(T >= 1 && E >= 1 && C >= 1, "T, E, C must be positive");
  CM3_REQUIRE(n_seg >= 1 && n_seg <= 4096, "n_segments must be in 1..4096");
  CopyShift cs;
  memset(&cs, 0, sizeof(cs));
  if (shift) {
    CM3_REQUIRE(shift->n >= 0 && shift->n <= 4, "copy shift: 0..4 regions");
    cs.n = shift->n;
    for (int r = 0; r < shift->n; ++r) {
      CM3_REQUIRE(shift->first_dst[r] && shift->mid[r] && shift->last_src[r], "copy shift: null region %d", r);
      CM3_REQUIRE(shift->bytes[r] % 16 == 0 && ((uintptr_t)shift->first_dst[r] % 16) == 0 && ((uintptr_t)shift->mid[r] % 16) == 0 &&
                      ((uintptr_t)shift->last_src[r] % 16) == 0,
                  "copy shift: region %d is not 16-byte aligned / sized", r);
      cs.first_dst[r] = shift->first_dst[r];
      cs.mid[r] = shift->mid[r];
      cs.last_src[r] = shift->last_src[r];
      cs.bytes[r] = shift->bytes[r];
    }
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t cols = (size_t)E * C;
  int blocks = (int)((cols + kAdvBlock - 1) / kAdvBlock);   // the column grid of k_returns_moments: the same fold, the same moments
  int n_partials;
  if (T <= kKeepTicks && blocks <= kAdvMaxBlocks) {
    n_partials = cs.n > 0 && blocks < kKeepCopyBlocks ? kKeepCopyBlocks : blocks;
    hipLaunchKernelGGL((k_returns_partials_keep<R, kKeepTicks>), dim3(n_partials, n_seg), dim3(kAdvBlock), 0, s, (const R *)x, done, valid,
                       (R *)out, (double *)scratch, T, E, C, (R)gamma, blocks, cs);
  } else {
    n_partials = blocks > kAdvMaxBlocks ? kAdvMaxBlocks : blocks;
    hipLaunchKernelGGL((k_returns_partials<R>), dim3(n_partials, n_seg), dim3(kAdvBlock), 0, s, (const R *)x, done, valid, (R *)out,
                       (double *)scratch, T, E, C, (R)gamma, cs);
  }
  CM3_HIP_CHECK(hipGetLastError());
  const size_t n_elem = (size_t)T * cols;
  size_t nblocks = apply ? (n_elem / 4 + kAdvBlock - 1) / kAdvBlock : 1;   // (four elements per lane: see k_fold_normalize)
  const size_t cap = n_seg > 1 ? (size_t)((2048 + n_seg - 1) / n_seg < 64 ? 64 : (2048 + n_seg - 1) / n_seg) : 2048;
  if (nblocks > cap) nblocks = cap;
  if (nblocks < 1) nblocks = 1;
  hipLaunchKernelGGL((k_fold_normalize<R>), dim3((unsigned)nblocks, n_seg), dim3(kAdvBlock), 0, s, (R *)out, valid,
                     (const double *)scratch, n_partials, moments, stats, n_elem, C, eps, apply);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

template <typename R>
static int returns_moments(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                           double *moments, int T, int E, int C, double gamma, void *stream) {
  CM3_REQUIRE(x && done && out && scratch && moments, "null pointer");
  CM3_REQUIRE(T >= 1 && E >= 1 && C >= 1, "T, E, C must be positive");
  const size_t cols = (size_t)E * C;
  int blocks = (int)((cols + kAdvBlock - 1) / kAdvBlock);
  if (blocks > kAdvMaxBlocks) blocks = kAdvMaxBlocks;
  hipStream_t s = (hipStream_t)stream;
  // The arrival counter must be zero when the launch starts: the caller zero-initialises `scratch` ONCE, and the block that
  // arrives last resets the counter for the next call.  (Round 2 zeroed it with a 4-byte memset per call -- one more dependent
  // launch, ~2 us of a 114 us C4 rollout; a launch that dies half-way takes the HIP context with it anyway.)
  hipLaunchKernelGGL((k_returns_moments<R>), dim3(blocks), dim3(kAdvBlock), 0, s, (const R *)x, done, valid, (R *)out,
                     (double *)scratch, moments, T, E, C, (R)gamma);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

template <typename R>
static int normalize(void *x, const uint8_t *valid, const double *parts, int n_parts, double *stats, size_t n_elem, int C,
                     double eps, int apply, void *stream, int n_seg = 1) {
  CM3_REQUIRE(parts && n_parts >= 1, "null moments / n_parts < 1");
  CM3_REQUIRE(n_seg >= 1 && n_seg <= 4096, "n_segments must be in 1..4096");
  CM3_REQUIRE(!apply || x, "null pointer");
  CM3_REQUIRE(n_elem >= 1 && C >= 1, "n_elem and C must be positive");
  size_t blocks = apply ? (n_elem + kAdvBlock - 1) / kAdvBlock : 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((k_normalize<R>), dim3((unsigned)blocks, n_seg), dim3(kAdvBlock), 0, (hipStream_t)stream, (R *)x, valid,
                     parts, n_parts, stats, n_elem, C, eps, apply);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

}  // namespace cm3

extern "C" {
size_t cm3_returns_scratch_bytes(void) { return (size_t)cm3::kAdvMaxBlocks * 3 * sizeof(double) + 16; }
int cm3_returns_moments_f32(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                            double *moments, int32_t T, int32_t E, int32_t C, double gamma, void *stream) {
  return cm3::returns_moments<float>(x, done, valid, out, scratch, moments, T, E, C, gamma, stream);
}
int cm3_returns_moments_f64(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                            double *moments, int32_t T, int32_t E, int32_t C, double gamma, void *stream) {
  return cm3::returns_moments<double>(x, done, valid, out, scratch, moments, T, E, C, gamma, stream);
}
int cm3_normalize_f32(void *x, const uint8_t *valid, const double *parts, int32_t n_parts, double *stats, size_t n_elem,
                      int32_t C, double eps, int32_t apply, void *stream) {
  return cm3::normalize<float>(x, valid, parts, n_parts, stats, n_elem, C, eps, apply, stream);
}
int cm3_normalize_f64(void *x, const uint8_t *valid, const double *parts, int32_t n_parts, double *stats, size_t n_elem,
                      int32_t C, double eps, int32_t apply, void *stream) {
  return cm3::normalize<double>(x, valid, parts, n_parts, stats, n_elem, C, eps, apply, stream);
}
int cm3_returns_normalize_f32(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                              double *moments, double *stats, int32_t T, int32_t E, int32_t C, double gamma, double eps,
                              int32_t apply, const cm3_copy_shift *shift, void *stream) {
  return cm3::returns_normalize<float>(x, done, valid, out, scratch, moments, stats, T, E, C, gamma, eps, apply, shift, stream);
}
int cm3_returns_normalize_f64(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                              double *moments, double *stats, int32_t T, int32_t E, int32_t C, double gamma, double eps,
                              int32_t apply, const cm3_copy_shift *shift, void *stream) {
  return cm3::returns_normalize<double>(x, done, valid, out, scratch, moments, stats, T, E, C, gamma, eps, apply, shift, stream);
}
size_t cm3_returns_segments_scratch_bytes(int32_t n_segments) {
  return (size_t)(n_segments < 1 ? 1 : n_segments) * cm3::kAdvMaxBlocks * 3 * sizeof(double) + 16;
}
int cm3_returns_normalize_segments_f32(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                                       double *moments, double *stats, int32_t T, int32_t n_segments, int32_t E, int32_t C,
                                       double gamma, double eps, int32_t apply, const cm3_copy_shift *shift, void *stream) {
  return cm3::returns_normalize<float>(x, done, valid, out, scratch, moments, stats, T, E, C, gamma, eps, apply, shift, stream, n_segments);
}
int cm3_returns_normalize_segments_f64(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                                       double *moments, double *stats, int32_t T, int32_t n_segments, int32_t E, int32_t C,
                                       double gamma, double eps, int32_t apply, const cm3_copy_shift *shift, void *stream) {
  return cm3::returns_normalize<double>(x, done, valid, out, scratch, moments, stats, T, E, C, gamma, eps, apply, shift, stream, n_segments);
}
int cm3_normalize_segments_f32(void *x, const uint8_t *valid, const double *parts, int32_t n_parts, int32_t n_segments, double *stats,
                               size_t n_elem, int32_t C, double eps, int32_t apply, void *stream) {
  return cm3::normalize<float>(x, valid, parts, n_parts, stats, n_elem, C, eps, apply, stream, n_segments);
}
int cm3_normalize_segments_f64(void *x, const uint8_t *valid, const double *parts, int32_t n_parts, int32_t n_segments, double *stats,
                               size_t n_elem, int32_t C, double eps, int32_t apply, void *stream) {
  return cm3::normalize<double>(x, valid, parts, n_parts, stats, n_elem, C, eps, apply, stream, n_segments);
}
int cm3_copy_list(int32_t n, void *const *dst, const void *const *src, const size_t *bytes, void *stream) {
  using namespace cm3;
  CM3_REQUIRE(n >= 1 && n <= 8 && dst && src && bytes, "copy list: 1..8 regions");
  CopyList c;
  size_t most = 0;
  c.n = n;
  for (int r = 0; r < n; ++r) {
    CM3_REQUIRE(dst[r] && src[r], "copy list: null region %d", r);
    CM3_REQUIRE(bytes[r] % 16 == 0 && ((uintptr_t)dst[r] % 16) == 0 && ((uintptr_t)src[r] % 16) == 0,
                "copy list: region %d is not 16-byte aligned / sized", r);
    c.dst[r] = dst[r];
    c.src[r] = src[r];
    c.bytes[r] = bytes[r];
    most = bytes[r] > most ? bytes[r] : most;
  }
  size_t blocks = (most / 16 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(k_copy_list, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, c);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}
}

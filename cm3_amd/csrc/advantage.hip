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
template <typename R>
__global__ void __launch_bounds__(kAdvBlock) k_returns_moments(const R *__restrict__ x, const uint8_t *__restrict__ done,
                                                               const uint8_t *__restrict__ valid, R *__restrict__ out,
                                                               double *__restrict__ partials, int T, int E, int C, R gamma) {
  const size_t cols = (size_t)E * C;
  double s = 0.0, s2 = 0.0, n = 0.0;
  for (size_t col = (size_t)blockIdx.x * kAdvBlock + threadIdx.x; col < cols; col += (size_t)gridDim.x * kAdvBlock) {
    const size_t e = col / C;
    R g = R(0);
    for (int t = T - 1; t >= 0; --t) {
      const size_t idx = (size_t)t * cols + col;
      const bool d = done[(size_t)t * E + e] != 0;
      g = x[idx] + (d ? R(0) : gamma * g);
      const bool v = valid ? (valid[(size_t)t * E + e] != 0) : true;
      out[idx] = v ? g : R(0);
      if (v) {
        const double gd = (double)g;
        s += gd;
        s2 += gd * gd;
        n += 1.0;
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

// one wave folds the per-block partials in a fixed order
__global__ void __launch_bounds__(64) k_fold_partials(const double *__restrict__ partials, int n_blocks,
                                                      double *__restrict__ moments) {
  double s = 0.0, s2 = 0.0, n = 0.0;
  for (int b = threadIdx.x; b < n_blocks; b += 64) {
    s += partials[3 * b + 0];
    s2 += partials[3 * b + 1];
    n += partials[3 * b + 2];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    s2 += __shfl_down(s2, off, 64);
    n += __shfl_down(n, off, 64);
  }
  if (threadIdx.x == 0) {
    moments[0] = s;
    moments[1] = s2;
    moments[2] = n;
  }
}

template <typename R>
__global__ void __launch_bounds__(kAdvBlock) k_normalize(R *__restrict__ x, const uint8_t *__restrict__ valid,
                                                         const double *__restrict__ moments, size_t n_elem, int C, double eps) {
  const double cnt = moments[2] > 1.0 ? moments[2] : 1.0;
  const double mean = moments[0] / cnt;
  double var = moments[1] / cnt - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const R m = (R)mean, inv = (R)(1.0 / (::sqrt(var) + eps));
  for (size_t i = (size_t)blockIdx.x * kAdvBlock + threadIdx.x; i < n_elem; i += (size_t)gridDim.x * kAdvBlock) {
    const bool v = valid ? (valid[i / C] != 0) : true;
    x[i] = v ? (x[i] - m) * inv : R(0);
  }
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
  hipLaunchKernelGGL((k_returns_moments<R>), dim3(blocks), dim3(kAdvBlock), 0, s, (const R *)x, done, valid, (R *)out,
                     (double *)scratch, T, E, C, (R)gamma);
  CM3_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_fold_partials, dim3(1), dim3(64), 0, s, (const double *)scratch, blocks, moments);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

template <typename R>
static int normalize(void *x, const uint8_t *valid, const double *moments, size_t n_elem, int C, double eps, void *stream) {
  CM3_REQUIRE(x && moments, "null pointer");
  CM3_REQUIRE(n_elem >= 1 && C >= 1, "n_elem and C must be positive");
  size_t blocks = (n_elem + kAdvBlock - 1) / kAdvBlock;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((k_normalize<R>), dim3((unsigned)blocks), dim3(kAdvBlock), 0, (hipStream_t)stream, (R *)x, valid,
                     moments, n_elem, C, eps);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

}  // namespace cm3

extern "C" {
size_t cm3_returns_scratch_bytes(void) { return (size_t)cm3::kAdvMaxBlocks * 3 * sizeof(double); }
int cm3_returns_moments_f32(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                            double *moments, int32_t T, int32_t E, int32_t C, double gamma, void *stream) {
  return cm3::returns_moments<float>(x, done, valid, out, scratch, moments, T, E, C, gamma, stream);
}
int cm3_returns_moments_f64(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                            double *moments, int32_t T, int32_t E, int32_t C, double gamma, void *stream) {
  return cm3::returns_moments<double>(x, done, valid, out, scratch, moments, T, E, C, gamma, stream);
}
int cm3_normalize_f32(void *x, const uint8_t *valid, const double *moments, size_t n_elem, int32_t C, double eps, void *stream) {
  return cm3::normalize<float>(x, valid, moments, n_elem, C, eps, stream);
}
int cm3_normalize_f64(void *x, const uint8_t *valid, const double *moments, size_t n_elem, int32_t C, double eps, void *stream) {
  return cm3::normalize<double>(x, valid, moments, n_elem, C, eps, stream);
}
}

// kernels_summary.hip -- posterior summaries on the device.
//
// The reference summarises an event from its .miso file (misopy/samples_utils.py:263-329 ->
// credible_intervals.py:4-72): posterior mean of every isoform's psi and the Chen-Shao interval,
// i.e. the order statistics number round(alpha/2 n) and round((1-alpha/2) n) of the sorted samples.
// Doing that where the samples already are avoids shipping 8(K+1)S bytes per event over PCIe
// (4.8 GB for the 40 000-event benchmark batch) just to reduce them to 3K numbers.
//
// One workgroup per (event, isoform) column:
//   * mean: thread t sums samples t, t+256, ... in order, then a fixed binary tree over the 256
//     partial sums -- a deterministic order the CPU checker reproduces exactly;
//   * order statistics: exact radix select (8 passes of 8 bits over order-preserving 64-bit keys),
//     no sort, no extra memory; the value returned IS one of the samples.
#include <hip/hip_runtime.h>

#include "device.hpp"

namespace miso {

__device__ __forceinline__ uint64_t order_key(double x) {
  const uint64_t u = __double_as_longlong(x);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(uint64_t k) {
  const uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double(u);
}

constexpr int SUMMARY_CACHE = 32;   // samples per thread held in registers (S <= 8192)

template <bool CACHED>
__device__ __forceinline__ void summarize_column(const double *x, int K, int S, int rank_lo, int rank_hi,
                                                 double *o) {
  __shared__ double part[256];
  __shared__ unsigned hist[256];
  __shared__ uint64_t s_prefix;
  __shared__ int s_rank;
  const int t = threadIdx.x;

  uint64_t keys[CACHED ? SUMMARY_CACHE : 1];
  double acc = 0.0;
  if (CACHED) {
#pragma unroll
    for (int j = 0; j < SUMMARY_CACHE; j++) {
      const int s = t + 256 * j;
      if (s < S) {
        const double v = x[static_cast<size_t>(s) * K];
        acc = acc + v;
        keys[j] = order_key(v);
      } else {
        keys[j] = 0;
      }
    }
  } else {
    for (int s = t; s < S; s += 256) acc = acc + x[static_cast<size_t>(s) * K];
  }
  part[t] = acc;
  __syncthreads();
  for (int stride = 128; stride >= 1; stride >>= 1) {
    if (t < stride) part[t] = part[t] + part[t + stride];
    __syncthreads();
  }
  const double mean = part[0] / static_cast<double>(S);

  double stat[2];
  for (int which = 0; which < 2; which++) {
    if (t == 0) { s_prefix = 0; s_rank = which == 0 ? rank_lo : rank_hi; }
    __syncthreads();
    for (int byte = 7; byte >= 0; byte--) {
      hist[t] = 0;
      __syncthreads();
      const uint64_t prefix = s_prefix;
      const uint64_t mask = (byte == 7) ? 0ull : (~0ull << (8 * (byte + 1)));
      if (CACHED) {
#pragma unroll
        for (int j = 0; j < SUMMARY_CACHE; j++) {
          if (t + 256 * j < S && (keys[j] & mask) == prefix)
            atomicAdd(&hist[(keys[j] >> (8 * byte)) & 0xFF], 1u);
        }
      } else {
        for (int s = t; s < S; s += 256) {
          const uint64_t key = order_key(x[static_cast<size_t>(s) * K]);
          if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * byte)) & 0xFF], 1u);
        }
      }
      __syncthreads();
      if (t == 0) {
        int r = s_rank, b = 0;
        while (b < 255 && r >= static_cast<int>(hist[b])) { r -= hist[b]; b++; }
        s_rank = r;
        s_prefix = prefix | (static_cast<uint64_t>(b) << (8 * byte));
      }
      __syncthreads();
    }
    stat[which] = key_value(s_prefix);
    __syncthreads();
  }
  if (t == 0) { o[0] = mean; o[1] = stat[0]; o[2] = stat[1]; }
}

__global__ __launch_bounds__(256) void summarize_kernel(const DevEvent *events, const unsigned char *out_pool,
                                                        int n_events, int S, int rank_lo, int rank_hi,
                                                        const uint64_t *sum_off, double *summary) {
  const int ev = blockIdx.x, k = blockIdx.y;
  if (ev >= n_events) return;
  const DevEvent E = events[ev];
  if (k >= E.K) return;
  const double *x = reinterpret_cast<const double *>(out_pool + E.off_samples) + k;
  double *o = summary + sum_off[ev] + 3 * k;
  if (S <= 256 * SUMMARY_CACHE) summarize_column<true>(x, E.K, S, rank_lo, rank_hi, o);
  else summarize_column<false>(x, E.K, S, rank_lo, rank_hi, o);
}

}  // namespace miso

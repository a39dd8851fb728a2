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
#include "miso_detmath.h"

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

// TEXT: summarise what summarize_miso READS, not what the sampler computed: the reference's summaries come from the
// `.miso` file (samples_utils.py:130-262 -> credible_intervals.py), where every sample is the text "%.4f" of psi.
// k = psi x 10^4 rounded to the nearest integer, ties to even, decided on the EXACT product (fma gives the rounding
// error of the multiplication), which is what a correctly rounded "%.4f" prints; the value read back is the double
// nearest to k / 10^4 = the correctly rounded quotient.  The mean is then the exact integer sum of the k over
// S x 10^4 (correctly rounded; numpy's pairwise float sum of the same values agrees to the last bit or two).
__device__ __forceinline__ long long text_digits(double v) {
  const double p = v * 10000.0;
  const double e = __builtin_fma(v, 10000.0, -p);        // v x 10^4 = p + e exactly
  double r = __builtin_rint(p);                          // nearest, ties to even
  const double d = (p - r) + e;                          // exact: both terms are tiny and on a common grid
  if (d > 0.5) r = r + 1.0; else if (d < -0.5) r = r - 1.0;
  else if (d == 0.5 && (static_cast<long long>(r) & 1)) r = r + 1.0;     // a tie decided against an odd r
  else if (d == -0.5 && (static_cast<long long>(r) & 1)) r = r - 1.0;
  return static_cast<long long>(r);
}

template <bool CACHED, bool TEXT>
__device__ __forceinline__ void summarize_column(const double *x, int K, int S, int rank_lo, int rank_hi,
                                                 double *o) {
  __shared__ double part[256];
  __shared__ long long ipart[256];
  __shared__ int s_nonfinite;
  if (TEXT) { if (threadIdx.x == 0) s_nonfinite = 0; __syncthreads(); }
  long long iacc = 0;
  auto as_read = [&](double v) {   // TEXT: the sample as the `.miso` file hands it on
    if (!TEXT) return v;
    if (!(v == v) || v - v != 0.0) { s_nonfinite = 1; return v; }     // "nan" / "inf" stay what they are
    const long long k = text_digits(v);
    iacc += k;
    return static_cast<double>(k) / 10000.0;
  };
  __shared__ unsigned hist2[2][256];
  __shared__ unsigned wave_tot2[2][4];
  __shared__ uint64_t s_prefix2[2];
  __shared__ int s_rank2[2];
  __shared__ int s_single2[2];
  const int t = threadIdx.x;

  uint64_t keys[CACHED ? SUMMARY_CACHE : 1];
  double acc = 0.0;
  if (CACHED) {
#pragma unroll
    for (int j = 0; j < SUMMARY_CACHE; j++) {
      const int s = t + 256 * j;
      if (s < S) {
        const double v = as_read(x[static_cast<size_t>(s) * K]);
        acc = acc + v;
        keys[j] = order_key(v);
      } else {
        keys[j] = 0;
      }
    }
  } else {
    for (int s = t; s < S; s += 256) acc = acc + as_read(x[static_cast<size_t>(s) * K]);
  }
  part[t] = acc;
  if (TEXT) ipart[t] = iacc;
  __syncthreads();
  for (int stride = 128; stride >= 1; stride >>= 1) {
    if (t < stride) { part[t] = part[t] + part[t + stride]; if (TEXT) ipart[t] = ipart[t] + ipart[t + stride]; }
    __syncthreads();
  }
  const double mean = !TEXT ? part[0] / static_cast<double>(S)
                            : (s_nonfinite ? part[0] / static_cast<double>(S)
                                           : static_cast<double>(ipart[0]) / (static_cast<double>(S) * 10000.0));

  // both order statistics in the same eight passes: two prefixes, two histograms (the keys are
  // tested against both; while the two ranks still share a prefix the two histograms are equal)
  double stat[2];
  if (t < 2) { s_prefix2[t] = 0; s_rank2[t] = t == 0 ? rank_lo : rank_hi; s_single2[t] = 0; }
  __syncthreads();
  for (int byte = 7; byte >= 0; byte--) {
    hist2[0][t] = 0; hist2[1][t] = 0;
    __syncthreads();
    const uint64_t p0 = s_prefix2[0], p1 = s_prefix2[1];
    const uint64_t mask = (byte == 7) ? 0ull : (~0ull << (8 * (byte + 1)));
    if (CACHED) {
      // psi lies in [0, 1]: the leading bytes of all keys coincide, so in the first passes every key
      // of the column lands in ONE bin and 5000 same-address LDS atomics serialise.  Each thread
      // therefore adds runs of equal bins at once, and a wavefront whose pending runs all name the
      // same bin adds their total with a single atomic.
      unsigned cb[2] = {0u, 0u}, cn[2] = {0u, 0u};
#pragma unroll
      for (int j = 0; j < SUMMARY_CACHE; j++) {
        if (t + 256 * j < S) {
          const uint64_t km = keys[j] & mask;
          const unsigned bin = static_cast<unsigned>(keys[j] >> (8 * byte)) & 0xFFu;
#pragma unroll
          for (int w = 0; w < 2; w++) {
            if (km == (w == 0 ? p0 : p1)) {
              if (cn[w] && cb[w] != bin) { atomicAdd(&hist2[w][cb[w]], cn[w]); cn[w] = 0; }
              cb[w] = bin; cn[w]++;
            }
          }
        }
      }
#pragma unroll
      for (int w = 0; w < 2; w++) {
        const unsigned long long have = __ballot(cn[w] != 0);
        if (have) {
          const unsigned b0 = static_cast<unsigned>(__shfl(static_cast<int>(cb[w]), __ffsll(static_cast<long long>(have)) - 1));
          if (__all(cn[w] == 0 || cb[w] == b0)) {
            unsigned tot = cn[w];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(static_cast<int>(tot), off);
            if ((t & 63) == 0) atomicAdd(&hist2[w][b0], tot);
          } else if (cn[w]) {
            atomicAdd(&hist2[w][cb[w]], cn[w]);
          }
        }
      }
    } else {
      for (int s = t; s < S; s += 256) {
        const double vq = x[static_cast<size_t>(s) * K];
        const uint64_t key = order_key((TEXT && vq == vq && vq - vq == 0.0) ? static_cast<double>(text_digits(vq)) / 10000.0 : vq);
        const uint64_t km = key & mask;
        const unsigned bin = static_cast<unsigned>(key >> (8 * byte)) & 0xFFu;
        if (km == p0) atomicAdd(&hist2[0][bin], 1u);
        if (km == p1) atomicAdd(&hist2[1][bin], 1u);
      }
    }
    __syncthreads();
    {
      // which bin holds the wanted rank: inclusive prefix sums of the 256 bins, one bin per thread
      // (wave scan + the earlier waves' totals through LDS) -- thread 0 walking the bins one LDS
      // read at a time was ~25 k cycles per pass
      unsigned h[2], inc[2];
#pragma unroll
      for (int w = 0; w < 2; w++) {
        h[w] = hist2[w][t]; inc[w] = h[w];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned v = __shfl_up(inc[w], off);
          if ((t & 63) >= off) inc[w] += v;
        }
        if ((t & 63) == 63) wave_tot2[w][t >> 6] = inc[w];
      }
      const int r0[2] = {s_rank2[0], s_rank2[1]};
      __syncthreads();
#pragma unroll
      for (int w = 0; w < 2; w++) {
        unsigned before = 0;
        for (int q = 0; q < (t >> 6); q++) before += wave_tot2[w][q];
        const unsigned incl = inc[w] + before, exc = incl - h[w];
        const unsigned r = static_cast<unsigned>(r0[w]);
        // the first bin whose running total exceeds the rank; the last bin takes what is left
        if ((r >= exc && r < incl) || (t == 255 && r >= incl)) {
          s_rank2[w] = static_cast<int>(r - exc);
          s_prefix2[w] = (w == 0 ? p0 : p1) | (static_cast<uint64_t>(t) << (8 * byte));
          s_single2[w] = h[w] == 1u;   // one key left under this prefix: it IS the order statistic
        }
      }
    }
    __syncthreads();
    // Early exit: 5000 keys rarely share more than their leading four or five bytes, so after that many
    // passes each wanted rank is alone in its bin; the remaining passes would only spell out its low bytes.
    // The lone key is fetched from the registers instead (the value returned is still one of the samples).
    if (CACHED && byte > 0 && s_single2[0] && s_single2[1]) {
      const uint64_t q0 = s_prefix2[0], q1 = s_prefix2[1];
      const uint64_t m2 = ~0ull << (8 * byte);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < SUMMARY_CACHE; j++) {
        if (t + 256 * j < S) {
          if ((keys[j] & m2) == q0) s_prefix2[0] = keys[j];
          if ((keys[j] & m2) == q1) s_prefix2[1] = keys[j];
        }
      }
      __syncthreads();
      break;
    }
  }
  stat[0] = key_value(s_prefix2[0]);
  stat[1] = key_value(s_prefix2[1]);
  if (t == 0) { o[0] = mean; o[1] = stat[0]; o[2] = stat[1]; }
}

__global__ __launch_bounds__(256) void summarize_kernel(const DevEvent *events, const unsigned char *out_pool,
                                                        int n_events, int S, int rank_lo, int rank_hi,
                                                        const uint64_t *sum_off, double *summary, int as_text) {
  const int ev = blockIdx.x, k = blockIdx.y;
  if (ev >= n_events) return;
  const DevEvent E = events[ev];
  if (k >= E.K) return;
  const double *x = reinterpret_cast<const double *>(out_pool + E.off_samples) + k;
  double *o = summary + sum_off[ev] + 3 * k;
  if (as_text) {
    if (S <= 256 * SUMMARY_CACHE) summarize_column<true, true>(x, E.K, S, rank_lo, rank_hi, o);
    else summarize_column<false, true>(x, E.K, S, rank_lo, rank_hi, o);
  } else if (S <= 256 * SUMMARY_CACHE) summarize_column<true, false>(x, E.K, S, rank_lo, rank_hi, o);
  else summarize_column<false, false>(x, E.K, S, rank_lo, rank_hi, o);
}


// ---- two-sample comparison --------------------------------------------
// compare_miso per (event, isoform) (misopy/hypothesis_test.py:89-179, 348-380): index-paired
// delta_s = psi1_s - psi2_s; if mean|delta| <= 0.009 or all deltas are identical the posterior is
// "null peaked" (density inf at 0 -> Bayes factor 0); otherwise a Gaussian kernel density estimate
// with covariance factor 0.3 (bandwidth^2 = 0.09 * unbiased variance, scipy.stats.gaussian_kde
// evaluated at 0) and Savage-Dickey BF = prior(0) / posterior(0) = 1 / posterior(0), 1e12 when the
// posterior density underflows to 0, capped at 1e12.  All sums use the fixed 256-strided + binary
// tree order of summarize_column so a CPU checker can reproduce them.
__device__ __forceinline__ double block_tree_sum(double v, double *part) {
  const int t = threadIdx.x;
  __syncthreads();
  part[t] = v;
  __syncthreads();
  for (int stride = 128; stride >= 1; stride >>= 1) {
    if (t < stride) part[t] = part[t] + part[t + stride];
    __syncthreads();
  }
  return part[0];
}

// four sums at once, each in block_tree_sum's order (same bits), one set of barriers
__device__ __forceinline__ void block_tree_sum4(double (&v)[4], double (*part4)[256]) {
  const int t = threadIdx.x;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; q++) part4[q][t] = v[q];
  __syncthreads();
  for (int stride = 128; stride >= 1; stride >>= 1) {
    if (t < stride) {
#pragma unroll
      for (int q = 0; q < 4; q++) part4[q][t] = part4[q][t] + part4[q][t + stride];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; q++) v[q] = part4[q][0];
}

// CACHED (S <= 256 * SUMMARY_CACHE): the paired differences stay in registers, so the two sample
// columns are read once instead of three times.
template <bool CACHED>
__device__ __forceinline__ void compare_column(const double *x1, const double *x2, int K, int S,
                                               double smoothing, double *o) {
  __shared__ double part4[4][256];
  __shared__ int s_diff;
  const int t = threadIdx.x;
  double *part = part4[0];
  const double n = static_cast<double>(S);

  if (t == 0) s_diff = 0;
  const double d0 = x1[0] - x2[0];
  double acc[4] = {0, 0, 0, 0};   // sum psi1, sum psi2, sum d, sum |d|
  double dv[CACHED ? SUMMARY_CACHE : 1];
  int differs = 0;
  if (CACHED) {
#pragma unroll
    for (int j = 0; j < SUMMARY_CACHE; j++) {
      const int s = t + 256 * j;
      dv[j] = 0.0;
      if (s < S) {
        const double u = x1[static_cast<size_t>(s) * K], v = x2[static_cast<size_t>(s) * K];
        const double d = u - v;
        acc[0] = acc[0] + u; acc[1] = acc[1] + v; acc[2] = acc[2] + d; acc[3] = acc[3] + fabs(d);
        differs |= (d - d0 != 0.0);
        dv[j] = d;
      }
    }
  } else {
    for (int s = t; s < S; s += 256) {
      const double u = x1[static_cast<size_t>(s) * K], v = x2[static_cast<size_t>(s) * K];
      const double d = u - v;
      acc[0] = acc[0] + u; acc[1] = acc[1] + v; acc[2] = acc[2] + d; acc[3] = acc[3] + fabs(d);
      differs |= (d - d0 != 0.0);
    }
  }
  block_tree_sum4(acc, part4);
  const double sum1 = acc[0], sum2 = acc[1], sumd = acc[2], sumabs = acc[3];
  if (differs) atomicOr(&s_diff, 1);
  __syncthreads();
  const bool all_same = s_diff == 0;
  const double mean_d = sumd / n, mad = sumabs / n;

  double bf, post = 0.0;
  if (mad <= 0.009 || all_same) {       // block-uniform: every thread sees the same sums
    bf = 0.0;
    post = __longlong_as_double(0x7FF0000000000000ull);
  } else {
    double av = 0;
    if (CACHED) {
#pragma unroll
      for (int j = 0; j < SUMMARY_CACHE; j++)
        if (t + 256 * j < S) { const double d = dv[j] - mean_d; av = av + d * d; }
    } else {
      for (int s = t; s < S; s += 256) {
        const double d = (x1[static_cast<size_t>(s) * K] - x2[static_cast<size_t>(s) * K]) - mean_d;
        av = av + d * d;
      }
    }
    const double var = block_tree_sum(av, part) / (n - 1.0);
    const double cov = var * (smoothing * smoothing);
    const double inv2 = 1.0 / (2.0 * cov);
    double ae = 0;
    if (CACHED) {
#pragma unroll
      for (int j = 0; j < SUMMARY_CACHE; j++)
        if (t + 256 * j < S) ae = ae + miso_det_exp(-(dv[j] * dv[j]) * inv2);
    } else {
      for (int s = t; s < S; s += 256) {
        const double d = x1[static_cast<size_t>(s) * K] - x2[static_cast<size_t>(s) * K];
        ae = ae + miso_det_exp(-(d * d) * inv2);
      }
    }
    const double se = block_tree_sum(ae, part);
    post = se / (n * miso_det_sqrt(6.283185307179586 * cov));
    if (post == 0.0) bf = 1e12;
    else { bf = 1.0 / post; if (bf > 1e12) bf = 1e12; }
  }
  if (t == 0) { o[0] = sum1 / n; o[1] = sum2 / n; o[2] = bf; o[3] = post; }
}

__global__ __launch_bounds__(256) void compare_kernel(const DevEvent *ev1, const unsigned char *pool1,
                                                      const DevEvent *ev2, const unsigned char *pool2,
                                                      int n_events, int S, double smoothing,
                                                      const uint64_t *cmp_off, double *out) {
  const int ev = blockIdx.x, k = blockIdx.y;
  if (ev >= n_events) return;
  const DevEvent E1 = ev1[ev], E2 = ev2[ev];
  const int K = E1.K;
  if (k >= K) return;
  const double *x1 = reinterpret_cast<const double *>(pool1 + E1.off_samples) + k;
  const double *x2 = reinterpret_cast<const double *>(pool2 + E2.off_samples) + k;
  double *o = out + cmp_off[ev] + 4 * k;
  if (S <= 256 * SUMMARY_CACHE) compare_column<true>(x1, x2, K, S, smoothing, o);
  else compare_column<false>(x1, x2, K, S, smoothing, o);
}

}  // namespace miso

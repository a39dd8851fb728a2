// detmath_n.hpp -- miso_detmath.h's exp / log for N arguments at once, coefficients in the caller's registers (device code).
#pragma once
#include "miso_detmath.h"

#pragma clang fp contract(off)

namespace miso {
namespace {

// ---- several arguments through one routine, statement by statement (round 5) ----
// A chain per lane is ONE dependency chain: with 625 wavefronts on 1024 SIMDs nothing else issues on a wavefront's
// SIMD, and the eleven transcendental routines of a Metropolis-Hastings step (miso.c:449-552) -- Horner chains of a dozen
// fused multiply-adds each -- ran one after the other at the latency of a dependent f64 operation, ~10 cycles per
// instruction (hipcc does not interleave two calls of miso_det_log: each reads its coefficient table behind a volatile
// barrier).  det_log_n / det_exp_n evaluate N arguments with the operations of miso_det_log / miso_det_exp
// (include/miso_detmath.h) in the same order PER ARGUMENT -- the same bits -- but step by step across the arguments, so
// that N independent chains are in flight.  The coefficients come from the CALLER's registers (kept in VGPRs for the whole
// run: sampler_lane<true>): miso_detmath.h reads them from constant memory at every call so that kernels with two or three
// wavefronts per SIMD keep their occupancy, but a wavefront alone on its SIMD then waits out a scalar-cache round trip
// two or three times per call -- 119 scalar loads and 78 waits in the kernel, most of them in the iteration loop.
template <int N> __device__ __forceinline__ void det_exp_n(const double (&x)[N], double (&out)[N], const double (&te)[12]) {
  const double LOG2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  double xm[N], kd[N], r[N], p[N];
  int k[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    xm[j] = (x[j] != x[j]) ? 0.0 : x[j];
    xm[j] = (xm[j] > 710.0) ? 710.0 : xm[j];
    xm[j] = (xm[j] < -746.0) ? -746.0 : xm[j];
    kd[j] = __builtin_floor(xm[j] * LOG2E + 0.5);
    k[j] = static_cast<int>(kd[j]);
    r[j] = miso_fma(-kd[j], LN2_HI, xm[j]);
    r[j] = miso_fma(-kd[j], LN2_LO, r[j]);
  }
#pragma unroll
  for (int j = 0; j < N; j++) p[j] = te[0];
#pragma unroll
  for (int i = 1; i < 12; i++) {
#pragma unroll
    for (int j = 0; j < N; j++) p[j] = miso_fma(p[j], r[j], te[i]);
  }
#pragma unroll
  for (int j = 0; j < N; j++) {
    p[j] = miso_fma(p[j], r[j], 1.0);
    p[j] = miso_fma(p[j], r[j], 1.0);
    const int k1 = k[j] / 2, k2 = k[j] - k1;
    double res = p[j] * miso_pow2i(k1) * miso_pow2i(k2);
    res = (x[j] > 709.782712893384) ? miso_u2d(0x7FF0000000000000ull) : res;
    res = (x[j] < -745.2) ? 0.0 : res;
    res = (x[j] != x[j]) ? x[j] : res;
    out[j] = res;
  }
}

template <int N> __device__ __forceinline__ void det_log_n(const double (&x)[N], double (&out)[N], const double (&tl)[12]) {
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const double SQRT2 = 1.41421356237309504880;
  double f[N], s[N], z[N], q[N], ed[N];
#pragma unroll
  for (int j = 0; j < N; j++) {
    const int sub = (miso_d2u(x[j]) >> 52) == 0;
    const double xs = sub ? x[j] * 18014398509481984.0 : x[j];
    const uint64_t u = miso_d2u(xs);
    int e = static_cast<int>((u >> 52) & 0x7FF) - 1023 + (sub ? -54 : 0);
    double m = miso_u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
    const int big = m > SQRT2;
    m = big ? m * 0.5 : m;
    e += big ? 1 : 0;
    f[j] = m - 1.0;
    s[j] = f[j] / (2.0 + f[j]);
    z[j] = s[j] * s[j];
    ed[j] = static_cast<double>(e);
  }
#pragma unroll
  for (int j = 0; j < N; j++) q[j] = tl[0];
#pragma unroll
  for (int i = 1; i < 12; i++) {
#pragma unroll
    for (int j = 0; j < N; j++) q[j] = miso_fma(q[j], z[j], tl[i]);
  }
#pragma unroll
  for (int j = 0; j < N; j++) {
    const double R = z[j] * q[j];
    double res = miso_fma(ed[j], LN2_HI, f[j] - (s[j] * (f[j] - R) - ed[j] * LN2_LO));
    res = (miso_d2u(x[j]) == 0x7FF0000000000000ull) ? x[j] : res;
    res = (x[j] == 0.0) ? miso_u2d(0xFFF0000000000000ull) : res;
    res = (x[j] < 0.0) ? miso_u2d(0x7FF8000000000000ull) : res;
    res = (x[j] != x[j]) ? x[j] : res;
    out[j] = res;
  }
}

// miso_det_sqrt for a positive, normal, finite argument (the only kind the binomial's set-up has: n r q >= 5): the same
// operations without the special cases' branches, so that it sits in one basic block with what runs beside it
__device__ __forceinline__ double det_sqrt_pos(double x) {
  const uint64_t u = miso_d2u(x);
  int e = static_cast<int>(u >> 52) - 1023;
  const int odd = e & 1;
  e = (e - odd) / 2;
  const double m = miso_u2d((u & 0x000FFFFFFFFFFFFFull) | (static_cast<uint64_t>(1023 + odd) << 52));
  double y = 1.1547 - 0.1634 * m;
  y = y * (1.5 - 0.5 * m * y * y);
  y = y * (1.5 - 0.5 * m * y * y);
  y = y * (1.5 - 0.5 * m * y * y);
  y = y * (1.5 - 0.5 * m * y * y);
  y = y * (1.5 - 0.5 * m * y * y);
  double g = m * y;
  const double h = 0.5 * y;
  double d = miso_fma(-g, g, m);
  g = miso_fma(d, h, g);
  d = miso_fma(-g, g, m);
  g = miso_fma(d, h, g);
  return g * miso_pow2i(e);
}

// the two coefficient tables into VGPRs, pinned (the empty asm keeps the compiler from re-reading them from constant memory
// where they are used)
__device__ __forceinline__ void det_tables_to_registers(double (&te)[12], double (&tl)[12]) {
#pragma unroll
  for (int i = 0; i < 12; i++) {
    double ve = miso_tab_exp[i], vl = miso_tab_log[i];
    asm volatile("" : "+v"(ve), "+v"(vl));
    te[i] = ve; tl[i] = vl;
  }
}
__device__ __forceinline__ double det_exp_t(double x, const double (&te)[12]) {
  const double in[1] = {x};
  double o[1];
  det_exp_n<1>(in, o, te);
  return o[0];
}
__device__ __forceinline__ double det_log_t(double x, const double (&tl)[12]) {
  const double in[1] = {x};
  double o[1];
  det_log_n<1>(in, o, tl);
  return o[0];
}

}  // namespace
}  // namespace miso

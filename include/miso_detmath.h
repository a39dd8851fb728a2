/*
 * miso_detmath.h -- deterministic IEEE-754 double transcendental functions shared by the HIP
 * kernels (device), the host library and the CPU checker's counter mode.
 *
 * Why it exists: the sampler's per-iteration scalar math (reference miso.c:97-241, 449-552) uses
 * libm log/exp and R-mathlib's qnorm5 (random.c:1384).  glibc's libm and the GPU's ocml differ
 * in the last ulp, and a one-ulp difference in psi would make "bit-exact assignment counts at a
 * fixed seed" a matter of luck.  These versions use only +,-,*,/ and fma(), each of which is
 * correctly rounded on x86-64 and on gfx950, so host and device produce identical bits by
 * construction.  Every file that includes this header MUST be compiled with -ffp-contract=off
 * (the only fused operations are the explicit fma() calls below).
 *
 * Accuracy (checked in tests/test_detmath.py against libm over 10^6 points each):
 * exp, log <= 2 ulp; sqrt <= 1 ulp; qnorm as AS241 (about 1e-16 relative).
 *
 *   miso_det_exp   range reduction x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor (Horner, fma)
 *   miso_det_log   x = 2^e m, m in [sqrt(1/2), sqrt(2)), s = f/(2+f), atanh series in s^2
 *   miso_det_sqrt  bit-trick seed for 1/sqrt, Newton iterations in fma arithmetic
 *   miso_det_qnorm Wichura, "Algorithm AS 241: The percentage points of the normal
 *                  distribution", Appl. Statist. 37 (1988) 477-484, routine PPND16 -- the same
 *                  algorithm the reference's splicing_qnorm5 (random.c:1384-1470) implements,
 *                  restated from the paper's coefficient tables.
 */
#ifndef MISO_DETMATH_H
#define MISO_DETMATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define MISO_DM __host__ __device__ __forceinline__
#else
#define MISO_DM static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

MISO_DM uint64_t miso_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
MISO_DM double miso_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
MISO_DM double miso_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

/* 2^n for n in [-1022, 1023] */
MISO_DM double miso_pow2i(int n) { return miso_u2d((uint64_t) (n + 1023) << 52); }

MISO_DM double miso_det_exp(double x) {
  const double LOG2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01; /* low 21 bits of the mantissa zero */
  const double LN2_LO = 1.90821492927058770002e-10;
  double kd, r, p;
  int k, k1, k2;
  if (x != x) return x;
  if (x > 709.782712893384) return miso_u2d(0x7FF0000000000000ull);
  if (x < -745.2) return 0.0;
  kd = __builtin_floor(x * LOG2E + 0.5);
  k = (int) kd;
  r = miso_fma(-kd, LN2_HI, x);
  r = miso_fma(-kd, LN2_LO, r);
  p = 1.0 / 6227020800.0;               /* 1/13! */
  p = miso_fma(p, r, 1.0 / 479001600.0); /* 1/12! */
  p = miso_fma(p, r, 1.0 / 39916800.0);
  p = miso_fma(p, r, 1.0 / 3628800.0);
  p = miso_fma(p, r, 1.0 / 362880.0);
  p = miso_fma(p, r, 1.0 / 40320.0);
  p = miso_fma(p, r, 1.0 / 5040.0);
  p = miso_fma(p, r, 1.0 / 720.0);
  p = miso_fma(p, r, 1.0 / 120.0);
  p = miso_fma(p, r, 1.0 / 24.0);
  p = miso_fma(p, r, 1.0 / 6.0);
  p = miso_fma(p, r, 0.5);
  p = miso_fma(p, r, 1.0);
  p = miso_fma(p, r, 1.0);
  k1 = k / 2;
  k2 = k - k1;
  return p * miso_pow2i(k1) * miso_pow2i(k2);
}

MISO_DM double miso_det_log(double x) {
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const double SQRT2 = 1.41421356237309504880;
  uint64_t u;
  int e;
  double m, f, s, z, q, R, ed;
  if (x != x) return x;
  if (x < 0.0) return miso_u2d(0x7FF8000000000000ull);
  if (x == 0.0) return miso_u2d(0xFFF0000000000000ull);
  u = miso_d2u(x);
  if (u == 0x7FF0000000000000ull) return x;
  e = 0;
  if ((u >> 52) == 0) { /* subnormal: scale by 2^54 */
    x = x * 18014398509481984.0;
    u = miso_d2u(x);
    e = -54;
  }
  e += (int) (u >> 52) - 1023;
  m = miso_u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
  if (m > SQRT2) { m = m * 0.5; e += 1; }
  f = m - 1.0;
  s = f / (2.0 + f);
  z = s * s;
  q = 2.0 / 25.0;
  q = miso_fma(q, z, 2.0 / 23.0);
  q = miso_fma(q, z, 2.0 / 21.0);
  q = miso_fma(q, z, 2.0 / 19.0);
  q = miso_fma(q, z, 2.0 / 17.0);
  q = miso_fma(q, z, 2.0 / 15.0);
  q = miso_fma(q, z, 2.0 / 13.0);
  q = miso_fma(q, z, 2.0 / 11.0);
  q = miso_fma(q, z, 2.0 / 9.0);
  q = miso_fma(q, z, 2.0 / 7.0);
  q = miso_fma(q, z, 2.0 / 5.0);
  q = miso_fma(q, z, 2.0 / 3.0);
  R = z * q;                      /* log(1+f) = 2s + s*R = f - s*(f - R) */
  ed = (double) e;
  return miso_fma(ed, LN2_HI, f - (s * (f - R) - ed * LN2_LO));
}

MISO_DM double miso_det_sqrt(double x) {
  uint64_t u;
  int e, odd;
  double m, y, h, g, d;
  if (x != x || x < 0.0) return miso_u2d(0x7FF8000000000000ull);
  if (x == 0.0) return x;
  u = miso_d2u(x);
  if (u == 0x7FF0000000000000ull) return x;
  e = 0;
  if ((u >> 52) == 0) { x = x * 18014398509481984.0; u = miso_d2u(x); e = -54; }
  e += (int) (u >> 52) - 1023;
  odd = e & 1;                       /* works for negative e too (two's complement) */
  e = (e - odd) / 2;
  /* m in [1,4) */
  m = miso_u2d((u & 0x000FFFFFFFFFFFFFull) | ((uint64_t) (1023 + odd) << 52));
  /* seed for 1/sqrt(m): linear fit on [1,4), refined by Newton */
  y = 1.1547 - 0.1634 * m;           /* crude: exact at neither end, error < 20 % */
  y = y * (1.5 - 0.5 * m * y * y);
  y = y * (1.5 - 0.5 * m * y * y);
  y = y * (1.5 - 0.5 * m * y * y);
  y = y * (1.5 - 0.5 * m * y * y);
  y = y * (1.5 - 0.5 * m * y * y);
  g = m * y;                         /* ~ sqrt(m) */
  h = 0.5 * y;
  d = miso_fma(-g, g, m);            /* residual m - g^2 (exact to one rounding) */
  g = miso_fma(d, h, g);
  d = miso_fma(-g, g, m);
  g = miso_fma(d, h, g);
  return g * miso_pow2i(e);
}

/* AS241 PPND16: lower-tail standard normal quantile. p in (0,1); p == 0 -> -inf, p == 1 -> +inf
   (as splicing_qnorm5, random.c:1392-1393). */
MISO_DM double miso_det_qnorm(double p) {
  double q, r, val;
  if (p != p) return p;
  if (p <= 0.0) return (p == 0.0) ? miso_u2d(0xFFF0000000000000ull) : miso_u2d(0x7FF8000000000000ull);
  if (p >= 1.0) return (p == 1.0) ? miso_u2d(0x7FF0000000000000ull) : miso_u2d(0x7FF8000000000000ull);
  q = p - 0.5;
  if ((q < 0 ? -q : q) <= 0.425) {
    r = 0.180625 - q * q;
    val = q * (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r +
                    67265.770927008700853) * r + 45921.953931549871457) * r +
                  13731.693765509461125) * r + 1971.5909503065514427) * r +
                133.14166789178437745) * r + 3.387132872796366608) /
          (((((((r * 5226.495278852854561 + 28729.085735721942674) * r +
                39307.89580009271061) * r + 21213.794301586595867) * r +
              5394.1960214247511077) * r + 687.1870074920579083) * r +
            42.313330701600911252) * r + 1.0);
    return val;
  }
  r = (q > 0) ? (1.0 - p) : p;
  r = miso_det_sqrt(-miso_det_log(r));
  if (r <= 5.0) {
    r = r - 1.6;
    val = (((((((r * 7.7454501427834140764e-4 + 0.0227238449892691845833) * r +
                0.24178072517745061177) * r + 1.27045825245236838258) * r +
              3.64784832476320460504) * r + 5.7694972214606914055) * r +
            4.6303378461565452959) * r + 1.42343711074968357734) /
          (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r +
                0.0151986665636164571966) * r + 0.14810397642748007459) * r +
              0.68976733498510000455) * r + 1.6763848301838038494) * r +
            2.05319162663775882187) * r + 1.0);
  } else {
    r = r - 5.0;
    val = (((((((r * 2.01033439929228813265e-7 + 2.71155556874348757815e-5) * r +
                0.0012426609473880784386) * r + 0.026532189526576123093) * r +
              0.29656057182850489123) * r + 1.7848265399172913358) * r +
            5.4637849111641143699) * r + 6.6579046435011037772) /
          (((((((r * 2.04426310338993978564e-15 + 1.4215117583164458887e-7) * r +
                1.8463183175100546818e-5) * r + 7.868691311456132591e-4) * r +
              0.0148753612908506148525) * r + 0.13692988092273580531) * r +
            0.59983220655588793769) * r + 1.0);
  }
  return (q < 0.0) ? -val : val;
}

/* The reference's normal variate (random.c:1543-1551): 59-bit uniform from two draws, then
   inversion.  u1, u2 are [0,1) uniforms. */
MISO_DM double miso_det_norm_from_unif(double u1, double u2) {
  const double BIG = 134217728.0; /* 2^27 */
  double u = (double) (int) (BIG * u1) + u2;
  return miso_det_qnorm(u / BIG);
}

#endif /* MISO_DETMATH_H */

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
#include <hip/hip_runtime.h>
#define MISO_DM __host__ __device__ __forceinline__
#else
#define MISO_DM static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

/* Polynomial coefficients live in tables: on the device they sit in constant memory (scalar
   loads into SGPRs when needed) instead of being hoisted into ~150 VGPRs by the compiler, which
   would halve the occupancy of every kernel that calls these routines in a loop.  The tables are
   not `const` on the device on purpose (a const table is folded back into immediates). */
#if defined(__HIP_DEVICE_COMPILE__)
#define MISO_TAB static __constant__ double
#else
#define MISO_TAB static const double
#endif

/* Device code reads a table through MISO_TAB_REF: the pointer passes through an empty volatile
   asm, so the compiler can neither fold the loads nor hoist them out of the sampler's iteration
   loop -- hoisting all 72 coefficients at once spills SGPRs into VGPR lanes, and the reloads land
   in the hot read loop.  Scalar loads from the constant cache at each use are far cheaper. */
#if defined(__HIP_DEVICE_COMPILE__)
typedef const double __attribute__((address_space(4))) *miso_ctab_t;
static __device__ __forceinline__ miso_ctab_t miso_tab_fresh(miso_ctab_t p) {
  __asm__ volatile("" : "+s"(p));
  return p;
}
#define MISO_TAB_REF(t) miso_tab_fresh((miso_ctab_t) (t))
#define MISO_TAB_PTR miso_ctab_t
#else
#define MISO_TAB_REF(t) (t)
#define MISO_TAB_PTR const double *
#endif

MISO_TAB miso_tab_exp[12] = { /* 1/13! ... 1/2! */
  1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0,
  1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5 };
MISO_TAB miso_tab_log[12] = { /* 2/25 ... 2/3 */
  2.0 / 25.0, 2.0 / 23.0, 2.0 / 21.0, 2.0 / 19.0, 2.0 / 17.0, 2.0 / 15.0, 2.0 / 13.0, 2.0 / 11.0,
  2.0 / 9.0, 2.0 / 7.0, 2.0 / 5.0, 2.0 / 3.0 };
/* AS241 PPND16 coefficient sets, highest degree first: a/b central, c/d intermediate, e/f tail */
MISO_TAB miso_tab_qa[8] = { 2509.0809287301226727, 33430.575583588128105, 67265.770927008700853,
  45921.953931549871457, 13731.693765509461125, 1971.5909503065514427, 133.14166789178437745,
  3.387132872796366608 };
MISO_TAB miso_tab_qb[8] = { 5226.495278852854561, 28729.085735721942674, 39307.89580009271061,
  21213.794301586595867, 5394.1960214247511077, 687.1870074920579083, 42.313330701600911252, 1.0 };
MISO_TAB miso_tab_qc[8] = { 7.7454501427834140764e-4, 0.0227238449892691845833,
  0.24178072517745061177, 1.27045825245236838258, 3.64784832476320460504, 5.7694972214606914055,
  4.6303378461565452959, 1.42343711074968357734 };
MISO_TAB miso_tab_qd[8] = { 1.05075007164441684324e-9, 5.475938084995344946e-4,
  0.0151986665636164571966, 0.14810397642748007459, 0.68976733498510000455,
  1.6763848301838038494, 2.05319162663775882187, 1.0 };
MISO_TAB miso_tab_qe[8] = { 2.01033439929228813265e-7, 2.71155556874348757815e-5,
  0.0012426609473880784386, 0.026532189526576123093, 0.29656057182850489123,
  1.7848265399172913358, 5.4637849111641143699, 6.6579046435011037772 };
MISO_TAB miso_tab_qf[8] = { 2.04426310338993978564e-15, 1.4215117583164458887e-7,
  1.8463183175100546818e-5, 7.868691311456132591e-4, 0.0148753612908506148525,
  0.13692988092273580531, 0.59983220655588793769, 1.0 };

MISO_DM uint64_t miso_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
MISO_DM double miso_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
MISO_DM double miso_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
/* fma(a, b, c) with c a table coefficient.  Device: c is wave-uniform and sits in an SGPR pair (scalar load);
   written as the one instruction it is -- hipcc otherwise copies the coefficient into a VGPR pair first (two
   v_mov_b32) to use the two-operand v_fmac_f64: three issue slots per Horner step instead of one, 22 of them
   per exp / log call (a fifth of the Metropolis-Hastings step's VALU instructions).  Same fused operation. */
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double miso_fma_tab(double a, double b, double c) {
  double r;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
  return r;
}
#else
#define miso_fma_tab miso_fma
#endif

/* 2^n for n in [-1022, 1023] */
MISO_DM double miso_pow2i(int n) { return miso_u2d((uint64_t) (n + 1023) << 52); }

/* Both routines are written branch-free (special cases patched in by selects at the end): on the
   GPU a data-dependent early return splits the caller into many small basic blocks, and the long
   f64 dependency chains can then no longer be interleaved with independent work. */
MISO_DM double miso_det_exp(double x) {
  const double LOG2E = 1.4426950408889634074;
  const double LN2_HI = 6.93147180369123816490e-01; /* low 21 bits of the mantissa zero */
  const double LN2_LO = 1.90821492927058770002e-10;
  double xm, kd, r, p, res;
  int k, k1, k2, i;
  xm = (x != x) ? 0.0 : x;
  xm = (xm > 710.0) ? 710.0 : xm;
  xm = (xm < -746.0) ? -746.0 : xm;
  kd = __builtin_floor(xm * LOG2E + 0.5);
  k = (int) kd;
  r = miso_fma(-kd, LN2_HI, xm);
  r = miso_fma(-kd, LN2_LO, r);
  {
    MISO_TAB_PTR te = MISO_TAB_REF(miso_tab_exp);
    p = te[0];
    for (i = 1; i < 12; i++) p = miso_fma_tab(p, r, te[i]);
  }
  p = miso_fma(p, r, 1.0);
  p = miso_fma(p, r, 1.0);
  k1 = k / 2;
  k2 = k - k1;
  res = p * miso_pow2i(k1) * miso_pow2i(k2);
  res = (x > 709.782712893384) ? miso_u2d(0x7FF0000000000000ull) : res;
  res = (x < -745.2) ? 0.0 : res;
  res = (x != x) ? x : res;
  return res;
}

MISO_DM double miso_det_log(double x) {
  const double LN2_HI = 6.93147180369123816490e-01;
  const double LN2_LO = 1.90821492927058770002e-10;
  const double SQRT2 = 1.41421356237309504880;
  uint64_t u;
  int e, i, sub, big;
  double xs, m, f, s, z, q, R, ed, res;
  sub = (miso_d2u(x) >> 52) == 0; /* +subnormal (or +0): scale by 2^54 */
  xs = sub ? x * 18014398509481984.0 : x;
  u = miso_d2u(xs);
  e = (int) ((u >> 52) & 0x7FF) - 1023 + (sub ? -54 : 0);
  m = miso_u2d((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
  big = m > SQRT2;
  m = big ? m * 0.5 : m;
  e += big ? 1 : 0;
  f = m - 1.0;
  s = f / (2.0 + f);
  z = s * s;
  {
    MISO_TAB_PTR tl = MISO_TAB_REF(miso_tab_log);
    q = tl[0];
    for (i = 1; i < 12; i++) q = miso_fma_tab(q, z, tl[i]);
  }
  R = z * q;                      /* log(1+f) = 2s + s*R = f - s*(f - R) */
  ed = (double) e;
  res = miso_fma(ed, LN2_HI, f - (s * (f - R) - ed * LN2_LO));
  res = (miso_d2u(x) == 0x7FF0000000000000ull) ? x : res;
  res = (x == 0.0) ? miso_u2d(0xFFF0000000000000ull) : res;
  res = (x < 0.0) ? miso_u2d(0x7FF8000000000000ull) : res;
  res = (x != x) ? x : res;
  return res;
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
  double q, r, val, num, den;
  int i;
  if (p != p) return p;
  if (p <= 0.0) return (p == 0.0) ? miso_u2d(0xFFF0000000000000ull) : miso_u2d(0x7FF8000000000000ull);
  if (p >= 1.0) return (p == 1.0) ? miso_u2d(0x7FF0000000000000ull) : miso_u2d(0x7FF8000000000000ull);
  q = p - 0.5;
  if ((q < 0 ? -q : q) <= 0.425) {
    r = 0.180625 - q * q;
    {
      MISO_TAB_PTR tn = MISO_TAB_REF(miso_tab_qa);
      MISO_TAB_PTR td = MISO_TAB_REF(miso_tab_qb);
      num = tn[0]; den = td[0];
      for (i = 1; i < 8; i++) { num = num * r + tn[i]; den = den * r + td[i]; }
    }
    return q * num / den;
  }
  r = (q > 0) ? (1.0 - p) : p;
  r = miso_det_sqrt(-miso_det_log(r));
  if (r <= 5.0) {
    r = r - 1.6;
    {
      MISO_TAB_PTR tn = MISO_TAB_REF(miso_tab_qc);
      MISO_TAB_PTR td = MISO_TAB_REF(miso_tab_qd);
      num = tn[0]; den = td[0];
      for (i = 1; i < 8; i++) { num = num * r + tn[i]; den = den * r + td[i]; }
    }
  } else {
    r = r - 5.0;
    {
      MISO_TAB_PTR tn = MISO_TAB_REF(miso_tab_qe);
      MISO_TAB_PTR td = MISO_TAB_REF(miso_tab_qf);
      num = tn[0]; den = td[0];
      for (i = 1; i < 8; i++) { num = num * r + tn[i]; den = den * r + td[i]; }
    }
  }
  val = num / den;
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

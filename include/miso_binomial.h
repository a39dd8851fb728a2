/*
 * miso_binomial.h -- exact binomial draws from the counter RNG: the arithmetic contract of the COLLAPSED Gibbs
 * step, shared by the HIP kernel (csrc/kernels_lane.hip) and the CPU checker (oracle/miso_oracle.c, collapsed mode).
 *
 * Why it exists: the reference reassigns every read by itself (miso.c:30-91 drawAssignment inside
 * miso.c:493-552; one uniform per read with >= 2 compatible isoforms).  For single-end data a read's
 * probabilities depend only on WHICH isoforms it is compatible with (miso.c:56-63: weights psi_k over the
 * compatible ones), and everything downstream uses the per-isoform COUNTS only (miso.c:243-307).  The reads of one
 * compatibility class are exchangeable, so the class's counts are Multinomial(n_class, psi restricted to the class),
 * which is drawn here as a chain of binomials: the Markov chain on (psi, counts) is the reference's, at O(classes x K)
 * instead of O(reads) per iteration.
 *
 * The sampler: inversion for n min(p, 1-p) < 10 (Kachitvichyanukul & Schmeiser's BINV), otherwise BTRS (Hoermann 1993:
 * transformed rejection; ~86 % of the trials end in its squeeze after one division), its acceptance test made against
 * the exact probabilities n! / (k! (n-k)!) r^k q^(n-k) through a table of log factorials rather than Stirling's series
 * (on a GPU all 64 lanes of a wavefront pay for the slowest lane's path: two table reads beat four logarithms).  Both
 * are exact up to the table's rounding (~1e-13 relative).  Uniforms come from a sequential word stream addressed like every other draw
 * (miso_philox.h): site MISO_SITE_COUNTS, words 0, 1, 2, ... of (seed, event, chain, iteration) in the order the draws
 * are made; transcendentals are miso_detmath.h's; no fused multiply-adds: the same bits on the host and on gfx950.
 */
#ifndef MISO_BINOMIAL_H
#define MISO_BINOMIAL_H

#include "miso_detmath.h"
#include "miso_philox.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#define MISO_SITE_COUNTS 3u

#if defined(__HIP_DEVICE_COMPILE__)
#define MISO_LF_PTR const double *__restrict__
#else
#define MISO_LF_PTR const double *
#endif
#define MISO_LF_AT(lf, i) ((lf)[(i)])

typedef struct {
  uint64_t seed;
  uint32_t event_id, chain, iteration, site, next;
  uint32_t w0, w1, w2, w3;   /* the block in use */
} miso_ustream;

MISO_HD void miso_ustream_init(miso_ustream *s, uint64_t seed, uint32_t event_id, uint32_t chain,
                               uint32_t iteration, uint32_t site) {
  s->seed = seed; s->event_id = event_id; s->chain = chain; s->iteration = iteration; s->site = site;
  s->next = 0; s->w0 = s->w1 = s->w2 = s->w3 = 0;
}

MISO_HD double miso_ustream_next(miso_ustream *s) {
  const uint32_t i = s->next & 3u;
  uint32_t w;
  if (i == 0) {
    const miso_u32x4 b = miso_draw_block(s->seed, s->event_id, s->chain, s->iteration, s->site, s->next >> 2);
    s->w0 = b.v[0]; s->w1 = b.v[1]; s->w2 = b.v[2]; s->w3 = b.v[3];
  }
  w = i == 0 ? s->w0 : (i == 1 ? s->w1 : (i == 2 ? s->w2 : s->w3));
  s->next++;
  return miso_u01(w);
}

/* n r < 10: sequential search from 0 (K&S's BINV), restarted with a fresh uniform if it runs past the bound.
   Round 5: the search WITHOUT its division per step.  BINV compares u_x = u - p_0 - ... - p_(x-1) with p_x and moves on by
   p_(x+1) = p_x (n - x) r / ((x + 1) q); here both sides carry the common factor B_x = prod_(i <= x) i q:
   U_x = u_x B_x, P_x = p_x B_x, so that  U_(x+1) = (U_x - P_x) (x + 1) q  and  P_(x+1) = P_x (n - x) r  -- the same test
   u_x > p_x (B_x > 0), four multiplications and a subtraction per step instead of a division (x stays below
   n r + 10 sqrt(n r q + 1) < 44, so B_x < 44! and nothing overflows).  On the GPU a wavefront that holds ONE chain in this
   regime runs these steps for all its lanes, and a division is twenty dependent instructions: such wavefronts set the
   collapsed launch's duration (profiles/r05_sampler_lane_ilp.txt). */
MISO_HD int32_t miso_binomial_inversion(miso_ustream *s, int32_t n, double r) {
  const double q = 1.0 - r;
  const double qn = miso_det_exp((double) n * miso_det_log(q));
  const double np = (double) n * r;
  double bound = np + 10.0 * miso_det_sqrt(np * q + 1.0);
  int32_t x = 0;
  double P = qn, U = miso_ustream_next(s);
  int guard = 0;
  if (bound > (double) n) bound = (double) n;
  while (U > P) {
    x++;
    if ((double) x > bound) {
      if (++guard > 64) return (int32_t) np;   /* cannot happen for finite inputs; never loop forever */
      x = 0; P = qn; U = miso_ustream_next(s);
    } else {
      U = (U - P) * ((double) x * q);
      P = P * ((double) (n - x + 1) * r);
    }
  }
  return x;
}

/* log(k!) for k = 0 .. n-1: the running sum of miso_det_log(k), compensated (Neumaier) so that the table is good to
   the last bits whatever n; built once per batch on the host (runtime.hip) and by the checker, with these operations */
MISO_HD void miso_logfact_fill(double *t, int32_t n) {
  double sum = 0.0, comp = 0.0;
  int32_t k;
  for (k = 0; k < n; k++) {
    if (k >= 2) {
      const double x = miso_det_log((double) k);
      const double s2 = sum + x;
      comp = comp + ((__builtin_fabs(sum) >= __builtin_fabs(x)) ? ((sum - s2) + x) : ((x - s2) + sum));
      sum = s2;
    }
    t[k] = sum + comp;
  }
}

/* n r >= 10, r <= 1/2: BTRS (Hoermann 1993, "The generation of binomial random variates", J. Stat. Comput. Simul. 46:
   transformed rejection with a squeeze), the acceptance test against the exact probabilities through the table of
   log factorials lf (at least n + 1 entries).  Trial t uses words 2t, 2t + 1 of the stream. */
MISO_HD int32_t miso_binomial_btrs(miso_ustream *s, int32_t n, double r, MISO_LF_PTR lf) {
  const double q = 1.0 - r, dn = (double) n;
  const double spq = miso_det_sqrt(dn * r * q);
  const double b = 1.15 + 2.53 * spq;
  const double a = -0.0873 + 0.0248 * b + 0.01 * r;
  const double c = dn * r + 0.5;
  const double vr = 0.92 - 4.2 / b;
  const double alpha = (2.83 + 5.1 / b) * spq;
  const double m = __builtin_floor((dn + 1.0) * r);
  const double lpq = miso_det_log(r / q);
  const double h = MISO_LF_AT(lf, (int32_t) m) + MISO_LF_AT(lf, n - (int32_t) m);
  int trial;
  for (trial = 0; trial < 4096; trial++) {
    const double u = miso_ustream_next(s) - 0.5;
    double v = miso_ustream_next(s);
    const double us = 0.5 - __builtin_fabs(u);
    const double k = __builtin_floor((2.0 * a / us + b) * u + c);
    if (!(k >= 0.0 && k <= dn)) continue;          /* (also us = 0) */
    if (us >= 0.07 && v <= vr) return (int32_t) k;  /* the squeeze: most trials end here */
    if (v == 0.0) return (int32_t) k;               /* log(0) = -inf passes every test */
    v = v * alpha / (a / (us * us) + b);
    if (miso_det_log(v) <= (h - MISO_LF_AT(lf, (int32_t) k) - MISO_LF_AT(lf, n - (int32_t) k)) + (k - m) * lpq) return (int32_t) k;
  }
  return (int32_t) m;   /* cannot happen for finite inputs (acceptance > 0.85 per trial); never loop forever */
}

/* Binomial(n, p).  p <= 0 or not a number: 0; p >= 1: n.  lf: log factorials, at least n + 1 entries. */
MISO_HD int32_t miso_binomial(miso_ustream *s, int32_t n, double p, MISO_LF_PTR lf) {
  double r;
  int32_t y;
  if (n <= 0 || !(p > 0.0)) return 0;
  if (p >= 1.0) return n;
  r = p > 0.5 ? 1.0 - p : p;
  y = ((double) n * r < 10.0) ? miso_binomial_inversion(s, n, r) : miso_binomial_btrs(s, n, r, lf);
  if (y < 0) y = 0;
  if (y > n) y = n;
  return p > 0.5 ? n - y : y;
}

#endif /* MISO_BINOMIAL_H */

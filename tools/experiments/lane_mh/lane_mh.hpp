// lane_mh.hpp -- the Metropolis-Hastings step of one chain on ONE lane: the reference's arithmetic written out serially
// (miso.c:449-471 propose + logit_inv, miso.c:97-122 proposal density, miso.c:124-163 / miso_paired.c:88-131 assignment
// score, miso.c:243-307 / miso_paired.c:133-174 joint score, miso.c:493-552 ratio, miso.c:869-880 accept), the terms of the
// CURRENT psi cached across iterations.  Same values, same summation orders, same RNG addresses as the flat passes of
// sampler_flat / sampler_grp and as the CPU checker's counter mode: bit-identical results.
//
// Who calls it: the one wavefront per workgroup that runs the scalar step for all the workgroup's chains (sampler_pel,
// kernels_pel.inl; sampler_flatl carries its own copy of the same code), 64 chains at a time, the other wavefronts waiting at
// a barrier.  The chain's vectors live in its LDS slice; slices are an odd number of 8-byte words apart, so the 64 lanes of
// an access never collide on a bank.
#pragma once
#include <hip/hip_runtime.h>

#include "miso_detmath.h"
#include "miso_philox.h"

#pragma clang fp contract(off)

namespace miso {

// byte offsets of the chain's vectors in its slice (the same for every lane: scalar registers); buffer 0 of psi / alpha /
// lp / tb / lr = the current state and its cached logs, buffer PR (entries) = the proposal
struct LaneMhOff { int al, psi, lp, tb, lr, tc, cst, isc, hm1, bas, cnt, st; };
// st: the chain's state block in its slice, LANE_MH_ST doubles: what the step keeps from one iteration to the next lives
// THERE, not in registers -- the wavefront that runs the step also runs read loops in between, whose register budget it
// would otherwise share for the whole run
enum { LST_JAC = 0, LST_LSE, LST_LGSUM, LST_LGEACH, LST_COVAR, LST_SD, LST_SIGMA, LST_HASH, LST_ACCEPTED, LANE_MH_ST };
// a view of one vector: base + offset, nothing kept in registers but the lane's slice address
template <typename T> struct LaneVec {
  unsigned char *p;
  __device__ __forceinline__ T &operator[](int k) const { return reinterpret_cast<T *>(p)[k]; }
};

struct LaneMh {
  unsigned char *mb;   // the lane's chain's slice
  LaneMhOff o;
  __device__ __forceinline__ LaneVec<double> D(int off) const { return LaneVec<double>{mb + off}; }
  __device__ __forceinline__ LaneVec<int> I(int off) const { return LaneVec<int>{mb + off}; }
  int PR;
  int lK;        // the chain's isoforms
  int Kmh;       // the wavefront's largest lK: scalar loop bound, lanes with fewer isoforms are switched off inside
  double lg_sum, lg_each, covar, sd, sigma;
  uint64_t seed; uint32_t evid, chain;
  // state across iterations
  double jac = 0.0, lse = 0.0;     // of the current psi
  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0;
};

// alpha' = alpha + sd z ; psi' = logit_inv(alpha'), then what both scores need of the new point and not of the counts:
// lp = log x, tb = lp + cst, lr = log(x_k / x_K'), the jacobian, the largest tb.  SRC / DST: buffer offsets (0 = current,
// PR = proposal) of alpha read / everything written.  With DST != 0 also the Gaussian exponents of the two proposal
// densities (miso.c:110-117): e1: current psi's log ratios against alpha', e2: the proposal's against alpha.
__device__ __forceinline__ void lane_mh_propose(const LaneMh &c, uint32_t iter, int SRC, int DST, double &jac_out,
                                                double &max_out, double &e1, double &e2, uint32_t &accw) {
  const LaneVec<double> al = c.D(c.o.al), psi = c.D(c.o.psi), lp = c.D(c.o.lp), tb = c.D(c.o.tb), lr = c.D(c.o.lr), tc = c.D(c.o.tc),
                        cst = c.D(c.o.cst);
  // Every loop handles TWO isoforms per trip, written out side by side: the routines are branch-free chains of dependent
  // f64 operations (~55 each), and one wavefront alone issues a dependent chain at a fraction of the SIMD's rate -- two
  // independent chains interleave.  Entries beyond the chain's isoforms are computed on clamped loads and dropped; the
  // sums still run in the reference's order.
  const int n1 = c.lK - 1;
  miso_u32x4 b = miso_draw_block(c.seed, c.evid, c.chain, iter, MISO_SITE_MH, 0u);
  accw = b.v[0];   // block 0, word 0 (miso.c:870)
  double acc = 0.0;
  for (int j = 0; j < c.Kmh - 1; j += 2) {
    // normal j uses words 2 + 2j, 3 + 2j of the site: even j -> words 2, 3 of block j / 2; odd j -> words 0, 1 of block (j + 1) / 2
    const miso_u32x4 bn = miso_draw_block(c.seed, c.evid, c.chain, iter, MISO_SITE_MH, static_cast<uint32_t>(j / 2 + 1));
    const int j1 = j + 1;
    const double z0 = miso_det_norm_from_unif(miso_u01(b.v[2]), miso_u01(b.v[3]));
    const double z1 = miso_det_norm_from_unif(miso_u01(bn.v[0]), miso_u01(bn.v[1]));
    const double an0 = al[SRC + j] + c.sd * z0, an1 = al[SRC + j1] + c.sd * z1;
    const double ex0 = miso_det_exp(an0), ex1 = miso_det_exp(an1);
    if (j < n1) { al[DST + j] = an0; tc[j] = ex0; acc = acc + ex0; }
    if (j1 < n1) { al[DST + j1] = an1; tc[j1] = ex1; acc = acc + ex1; }
    b = bn;
  }
  const double sumexp = acc + 1.0;
  double sumpsi = 0.0, ltheta = 1.0, prod = 1.0;
  for (int j = 0; j < c.Kmh - 1; j += 2) {
    const int j1 = j + 1;
    const double q0 = tc[j] / sumexp, q1 = tc[j1] / sumexp;
    if (j < n1) { psi[DST + j] = q0; sumpsi = sumpsi + q0; ltheta = ltheta - q0; prod = prod * q0; }
    if (j1 < n1) { psi[DST + j1] = q1; sumpsi = sumpsi + q1; ltheta = ltheta - q1; prod = prod * q1; }
  }
  psi[DST + n1] = 1 - sumpsi;
  jac_out = 1.0 / prod / ltheta;
  double maxv = 0.0;
  e1 = 0.0; e2 = 0.0;
  for (int k = 0; k < c.Kmh; k += 2) {
    const int k1 = k + 1;
    const double xv0 = psi[DST + k], xv1 = psi[DST + k1];
    const double ra = miso_det_log(xv0), rb = miso_det_log(xv1);
    const double r2a = miso_det_log(xv0 / ltheta), r2b = miso_det_log(xv1 / ltheta);
    const double ta = ra + cst[k], tb1 = rb + cst[k1];
    // the Gaussian exponents' terms (DST != 0): current psi's log ratios against alpha', the proposal's against alpha
    const double t1a = lr[k] - al[c.PR + k], t2a = r2a - al[k], t1b = lr[k1] - al[c.PR + k1], t2b = r2b - al[k1];
    const double g1a = (-0.5) * t1a * t1a / c.sigma, g2a = (-0.5) * t2a * t2a / c.sigma;
    const double g1b = (-0.5) * t1b * t1b / c.sigma, g2b = (-0.5) * t2b * t2b / c.sigma;
    if (k < c.lK) {
      lp[DST + k] = ra; tb[DST + k] = ta;
      maxv = (k == 0 || ta > maxv) ? ta : maxv;     // miso.c:137-140: maxv starts at entry 0
      if (k < n1) {
        if (DST != 0) { e1 = e1 + g1a; e2 = e2 + g2a; }
        lr[DST + k] = r2a;
      }
    }
    if (k1 < c.lK) {
      lp[DST + k1] = rb; tb[DST + k1] = tb1;
      maxv = (tb1 > maxv) ? tb1 : maxv;
      if (k1 < n1) {
        if (DST != 0) { e1 = e1 + g1b; e2 = e2 + g2b; }
        lr[DST + k1] = r2b;
      }
    }
  }
  max_out = maxv;
}

// sum_k exp(tb_k - maxv): the log-sum-exp's inner sum (miso.c:141-149)
__device__ __forceinline__ double lane_mh_sumexp(const LaneMh &c, int BUF, double maxv) {
  const LaneVec<double> tb = c.D(c.o.tb);
  double acc = 0.0;
  for (int k = 0; k < c.Kmh; k += 2) {   // two per trip: see lane_mh_propose
    const double x0 = miso_det_exp(tb[BUF + k] - maxv), x1 = miso_det_exp(tb[BUF + k + 1] - maxv);
    if (k < c.lK) acc = acc + x0;
    if (k + 1 < c.lK) acc = acc + x1;
  }
  return acc;
}

// state block <-> registers
__device__ __forceinline__ void lane_mh_begin(LaneMh &c) {
  const LaneVec<double> st = c.D(c.o.st);
  c.jac = st[LST_JAC]; c.lse = st[LST_LSE]; c.lg_sum = st[LST_LGSUM]; c.lg_each = st[LST_LGEACH]; c.covar = st[LST_COVAR];
  c.sd = st[LST_SD]; c.sigma = st[LST_SIGMA];
  c.hash = *reinterpret_cast<const uint64_t *>(c.mb + c.o.st + 8 * LST_HASH);
  c.accepted = *reinterpret_cast<const int *>(c.mb + c.o.st + 8 * LST_ACCEPTED);
}
__device__ __forceinline__ void lane_mh_end(const LaneMh &c) {
  const LaneVec<double> st = c.D(c.o.st);
  st[LST_JAC] = c.jac; st[LST_LSE] = c.lse;
  *reinterpret_cast<uint64_t *>(c.mb + c.o.st + 8 * LST_HASH) = c.hash;
  *reinterpret_cast<int *>(c.mb + c.o.st + 8 * LST_ACCEPTED) = c.accepted;
}
// the chain's constants (consts = the event's [3K ..]: lgamma(sum a), sum lgamma(a), sigma, sd, covar -- device.hpp) and its
// initial state: miso.c:834 (alpha + sd z in place), cached logs, log-sum-exp
__device__ __forceinline__ void lane_mh_init(LaneMh &c, const double *consts3k) {
  c.lg_sum = consts3k[0]; c.lg_each = consts3k[1]; c.sigma = consts3k[2]; c.sd = consts3k[3]; c.covar = consts3k[4];
  c.hash = 0xCBF29CE484222325ull; c.accepted = 0;
  double maxv, e1, e2; uint32_t accw;
  lane_mh_propose(c, MISO_ITER_INIT, 0, 0, c.jac, maxv, e1, e2, accw);
  c.lse = miso_det_log(lane_mh_sumexp(c, 0, maxv)) + maxv;
  const LaneVec<double> st = c.D(c.o.st);
  st[LST_LGSUM] = c.lg_sum; st[LST_LGEACH] = c.lg_each; st[LST_COVAR] = c.covar; st[LST_SD] = c.sd; st[LST_SIGMA] = c.sigma;
  lane_mh_end(c);
}

// One iteration's step for the counts in the slice.  PE: the read score is the caller's (rp_pe: the fixed-point sum of the
// picks' fragment scores, NaN when one of them is not finite -- miso_paired.c:157-163), the same for both points.
// Returns the log score to record (miso.c:882-893); the current state is updated on acceptance.
template <bool PE>
__device__ __forceinline__ double lane_mh_step(LaneMh &c, int m, double rp_pe) {
  lane_mh_begin(c);
  double jacN, maxN, e1, e2; uint32_t accw;
  lane_mh_propose(c, static_cast<uint32_t>(m), 0, c.PR, jacN, maxN, e1, e2, accw);
  const double sumtc = lane_mh_sumexp(c, c.PR, maxN);
  const double x1 = miso_det_exp(e1), x2 = miso_det_exp(e2);
  const double lseN = miso_det_log(sumtc) + maxN;
  const double ptoCS = miso_det_log(c.covar * c.jac * x1);    // miso.c:97-122: theta = psi,  mu = alpha'
  const double ctoPS = miso_det_log(c.covar * jacN * x2);     //                theta = psi', mu = alpha
  // joint log score of the proposal ([0]) and of the current point ([1]) for the current counts: rp / ap = the two
  // count-weighted sums, pq = the Dirichlet part; the counts' hash on the way
  double rp[2] = {0.0, 0.0}, ap[2] = {0.0, 0.0}, pq[2] = {0.0, 0.0};
  const int PR = c.PR;
  const LaneVec<double> al = c.D(c.o.al), psi = c.D(c.o.psi), lp = c.D(c.o.lp), tb = c.D(c.o.tb), lr = c.D(c.o.lr),
                        isc = c.D(c.o.isc), hm1 = c.D(c.o.hm1);
  const LaneVec<int> bas = c.I(c.o.bas), cnt = c.I(c.o.cnt);
  for (int k = 0; k < c.Kmh; k++) {
    if (k < c.lK) {
      const int cn = bas[k] + cnt[k];
      c.hash = (c.hash ^ static_cast<uint32_t>(cn)) * 0x100000001B3ull;
      const bool nz = cn != 0;
      const double ck = static_cast<double>(cn), hm = hm1[k];
      if (!PE) {
        const double is = isc[k];
        rp[0] = nz ? rp[0] + ck * is : rp[0];
        rp[1] = nz ? rp[1] + ck * is : rp[1];
      }
      ap[0] = nz ? ap[0] + ck * (tb[PR + k] - lseN) : ap[0];
      pq[0] = pq[0] + hm * lp[PR + k];
      ap[1] = nz ? ap[1] + ck * (tb[k] - c.lse) : ap[1];
      pq[1] = pq[1] + hm * lp[k];
    }
  }
  if (PE) { rp[0] = rp_pe; rp[1] = rp_pe; }
  double pj[2];
#pragma unroll
  for (int which = 0; which < 2; which++) {
    double psiProb = pq[which];
    psiProb = psiProb + c.lg_sum;
    psiProb = psiProb - c.lg_each;
    pj[which] = rp[which] + ap[which] + psiProb;
  }
  const double pp = pj[0], pc = pj[1];
  const double acceptP = (m > 0) ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);   // miso.c:493-552, 865
  const bool acc = (acceptP >= 1) || (miso_u01(accw) < acceptP);                                       // miso.c:869-880
  double cJS = pc;
  if (acc) {   // the proposal and its cached logs become the current state
    for (int k = 0; k < c.lK; k++) {
      psi[k] = psi[PR + k]; al[k] = al[PR + k]; lp[k] = lp[PR + k]; tb[k] = tb[PR + k]; lr[k] = lr[PR + k];
    }
    c.jac = jacN; c.lse = lseN; cJS = pp; c.accepted++;
  }
  lane_mh_end(c);
  return cJS;
}

}  // namespace miso

// kernels_lane.hip -- sampler_lane: two-isoform single-end events with the COLLAPSED Gibbs step, one chain per LANE.
//
// The reference reassigns every read by itself, one uniform each (miso.c:30-91 inside miso.c:493-552), and then uses
// only the per-isoform counts (miso.c:243-307).  Single-end reads compatible with both isoforms all have the same two
// probabilities (miso.c:56-63: psi_0 : psi_1), so their count on isoform 0 is Binomial(n_draw, psi_0 / (psi_0 + psi_1)):
// ONE exact binomial draw (include/miso_binomial.h: inversion / BTPE from the counter RNG) replaces the sweep over the
// reads -- the same Markov chain on (psi, counts), O(1) instead of O(reads) per iteration.  With the read sweep gone a
// chain is a few hundred scalar operations per iteration and needs no cooperation between lanes: every lane owns a
// chain (64 chains per wavefront instead of 16-21), state in registers, no LDS, no barrier.
//
// The run's LAST reassignment is made per read (Gibbs words of miso_philox.h, as sampler_k2 does): the assignment the
// caller gets back (miso.c:943-946) is then a per-read draw like the reference's.  Metropolis-Hastings step: the
// routines of kernels_k2.inl (psi_terms, joint, prop_exponent) called one after the other.
// Bit for bit against the CPU checker's collapsed mode (tests/test_gpu_collapsed.py).
#include "kernels_k2.inl"
#include "miso_binomial.h"
#include "detmath_n.hpp"

namespace miso {

#ifndef MISO_LANE_ILP
#define MISO_LANE_ILP 1
#endif

namespace {

// What miso_binomial_btrs (include/miso_binomial.h) computes before its first trial, for Binomial(n, p): kept for the
// chain's current psi and made for the proposal's BESIDE the Metropolis-Hastings scores (the proposal is accepted two
// times in three, and the wavefront has issue slots to spare), so that the accepted state's draw starts at its trials.
// log(k!): the head of the batch's table in LDS (sampler_lane_ilp stages it once), the rest in global memory.  Not for the
// bytes -- the table's hot entries sit in the L1 -- but for the COUNTER: a global load behind the iteration's sample stores
// waits for those stores to be acknowledged (vmcnt counts both, in order), a third of the average wavefront's time
// (SQ_WAIT_ANY, profiles/r05_sampler_lane_ilp.txt); an LDS read waits for nothing of the kind.
constexpr int LANE_LF_LDS = 2048;
struct LogFact {
  const double *__restrict__ glob;
  const double *lds;
  int n_lds;
  __device__ __forceinline__ double at(int32_t i) const { return i < n_lds ? lds[i] : glob[i]; }
};

struct BinomSetup {
  double p, r, spq, b, a, c, vr, alpha, m, lpq, h;
  bool btrs;      // n r >= 10 and 0 < p < 1: the fast path below; otherwise miso_binomial as it stands
};
// the part that needs no transcendental: everything but spq-dependent terms and lpq
__device__ __forceinline__ void binom_setup_pre(BinomSetup &s, int32_t n, double p, double &sq_arg, double &lg_arg) {
  s.p = p;
  const bool live = n > 0 && p > 0.0 && p < 1.0;
  s.r = p > 0.5 ? 1.0 - p : p;
  s.btrs = live && !(static_cast<double>(n) * s.r < 10.0);
  const double q = 1.0 - s.r, dn = static_cast<double>(n);
  sq_arg = s.btrs ? dn * s.r * q : 4.0;    // (a harmless argument where the fast path is not taken)
  lg_arg = s.btrs ? s.r / q : 1.0;
}
__device__ __forceinline__ void binom_setup_post(BinomSetup &s, int32_t n, double spq, double lpq, const LogFact &lf) {
  const double dn = static_cast<double>(n);
  s.spq = spq;
  s.b = 1.15 + 2.53 * spq;
  s.a = -0.0873 + 0.0248 * s.b + 0.01 * s.r;
  s.c = dn * s.r + 0.5;
  s.vr = 0.92 - 4.2 / s.b;
  s.alpha = (2.83 + 5.1 / s.b) * spq;
  s.m = __builtin_floor((dn + 1.0) * s.r);
  s.lpq = lpq;
  const int32_t mi = s.btrs ? static_cast<int32_t>(s.m) : 0;
  s.h = lf.at(mi) + lf.at(s.btrs ? n - mi : 0);
}
// the trials of miso_binomial_btrs, TWO per round: trials 2j and 2j + 1 are the two halves of block j of the chain's word
// stream (miso_ustream_next), evaluated side by side; the first that is accepted -- in the sequential order -- is the
// draw, exactly as the one-at-a-time loop returns it.  Every lane of the wavefront runs until its own draw is made; a
// lane that is not on the fast path (s.btrs false) takes no part.
__device__ __forceinline__ int32_t binom_trials(const BinomSetup &s, int32_t n, uint64_t seed, uint32_t event_id, uint32_t chain,
                                                uint32_t iter, const LogFact &lf, const double (&tl)[12]) {
  const double dn = static_cast<double>(n);
  int32_t y = static_cast<int32_t>(s.m);
  bool done = !s.btrs;
  for (uint32_t j = 0; j < 2048u && !done; j++) {
    const miso_u32x4 blk = miso_draw_block(seed, event_id, chain, iter, MISO_SITE_COUNTS, j);
    double u[2], v[2], us[2], k[2], lv[2], vv[2];
    bool in[2], quick[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      u[t] = miso_u01(blk.v[2 * t]) - 0.5;
      v[t] = miso_u01(blk.v[2 * t + 1]);
      us[t] = 0.5 - __builtin_fabs(u[t]);
      k[t] = __builtin_floor((2.0 * s.a / us[t] + s.b) * u[t] + s.c);
      in[t] = k[t] >= 0.0 && k[t] <= dn;
      quick[t] = (us[t] >= 0.07 && v[t] <= s.vr) || v[t] == 0.0;
      vv[t] = v[t] * s.alpha / (s.a / (us[t] * us[t]) + s.b);
    }
    det_log_n<2>(vv, lv, tl);
    bool ok[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const int32_t ki = in[t] ? static_cast<int32_t>(k[t]) : 0;
      const double rhs = (s.h - lf.at(ki) - lf.at(n - ki)) + (k[t] - s.m) * s.lpq;
      ok[t] = in[t] && (quick[t] || lv[t] <= rhs);
    }
    if (ok[0]) { y = static_cast<int32_t>(k[0]); done = true; }
    else if (ok[1]) { y = static_cast<int32_t>(k[1]); done = true; }
  }
  if (y < 0) y = 0;
  if (y > n) y = n;
  return s.p > 0.5 ? n - y : y;
}

}  // namespace

// (Tried: the head of the log-factorial table in LDS -- 60.5 ms against 61.2 ms from global memory with workgroups of one
// wavefront, 56.4 ms with these workgroups of four: the table's few hundred hot entries sit in the L1 anyway.)
// ILP (round 5): the form for batches that leave a wavefront alone, or nearly, on its SIMD (up to two per SIMD: 131 072
// chains) -- the step's transcendentals in five multi-argument stages, their coefficients in registers, the binomial's
// set-up beside the scores and its trials two per round; 250 registers.  The plain form (76 registers, six wavefronts per
// SIMD) for larger batches, where other wavefronts fill the gaps.  Same draws, same bits (tests/test_gpu_collapsed.py).
template <bool ILP>
__device__ __forceinline__ void lane_body(const KernelArgs &a, double *lds_lf) {
  LogFact LF{a.logfact, lds_lf, 0};
  if constexpr (ILP) {   // (a.tstride: the table's entries, runtime.hip)
    LF.n_lds = min(a.tstride, LANE_LF_LDS);
    for (int i = threadIdx.x; i < LF.n_lds; i += 256) lds_lf[i] = a.logfact[i];
    __syncthreads();   // the only barrier, before any thread leaves
  }
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  // a.pair_waves (runtime.hip): chains per wavefront when the batch has fewer wavefronts than the device has SIMDs -- a
  // wavefront costs what its slowest lane costs (the binomial's trials, the inversion's steps), fewer lanes are done sooner
  const int cpw = a.pair_waves > 0 ? a.pair_waves : 64;
  const int wl = threadIdx.x & 63;
  const long slot = (static_cast<long>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * cpw + wl;
  if (wl >= cpw || slot >= n_chains) return;   // no barrier below
  const int ev = a.slot_event[slot / a.C];
  const uint32_t chain = static_cast<uint32_t>(slot % a.C);
  const DevEvent E = a.events[ev];
  const uint32_t event_id = E.has_id ? E.explicit_id : a.first_event_id + static_cast<uint32_t>(ev);
  const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
  const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);
  K2Consts c;
  c.cst0 = consts[0]; c.cst1 = consts[1]; c.is0 = consts[2]; c.is1 = consts[3];
  c.hm0 = consts[4]; c.hm1 = consts[5]; c.lg_sum = consts[6]; c.lg_each = consts[7];
  c.sigma = consts[8]; c.sd = consts[9]; c.covar = consts[10];
  const int base0 = base[0], base1 = base[1];
  const int n_draw = E.n_draw;
  double *samples = reinterpret_cast<double *>(a.out_pool + E.off_samples);
  double *loglik = reinterpret_cast<double *>(a.out_pool + E.off_loglik);
  uint8_t *drawass = a.out_pool + E.off_drawass;
  int32_t *trace = (E.off_trace == NO_TRACE) ? nullptr : reinterpret_cast<int32_t *>(a.out_pool + E.off_trace);
  const uint32_t k0 = static_cast<uint32_t>(a.seed), k1 = static_cast<uint32_t>(a.seed >> 32);
  const uint32_t c2_gibbs = MISO_SITE_GIBBS | (chain << 8), c2_mh = MISO_SITE_MH | (chain << 8);

  PsiTerms cur;
  double alpha = 0.0;
  int cnt0 = 0, cnt1 = 0;

  // the counts of one Gibbs step, collapsed: x ~ Binomial(n_draw, psi_0 / S), S = (0 + psi_1) + psi_0
  // (the contract sums the class's psi from the last compatible isoform down)
  auto gibbs_collapsed = [&](uint32_t iter) {
    int d0 = 0;
    if (n_draw > 0) {
      miso_ustream us;
      miso_ustream_init(&us, a.seed, event_id, chain, iter, MISO_SITE_COUNTS);
      const double s = (0.0 + cur.x1) + cur.x0;
#ifdef MISO_LANE_NO_BINOM   // timing experiment: no draw
      d0 = static_cast<int>(static_cast<double>(n_draw) * (cur.x0 / s));
#else
      d0 = miso_binomial(&us, n_draw, cur.x0 / s, a.logfact);
#endif
    }
    cnt0 = base0 + d0;
    cnt1 = base1 + (n_draw - d0);
  };
  // ... and read by read (miso.c:69-73: U (psi_0 + psi_1) < psi_0 picks isoform 0), the r-th drawing read's uniform from
  // the Gibbs sites; chain 0's picks are what the caller gets back
  auto gibbs_per_read = [&](uint32_t iter) {
    const double p0 = 0.0 + cur.x0, T = p0 + cur.x1;
    // (a two-isoform read's uniform comes in two half-words, eight reads per block: include/miso_philox.h, lazy low bits)
    const int nb = (n_draw + 7) >> 3;
    int d0 = 0;
    for (int q = 0; q < nb; q++) {
      const miso_u32x4 hi = miso_philox4x32(static_cast<uint32_t>(q), iter, c2_gibbs, event_id, k0, k1);
      const miso_u32x4 lo = miso_philox4x32(static_cast<uint32_t>(q), iter, MISO_SITE_GIBBS_LOW | (chain << 8), event_id, k0, k1);
#pragma unroll
      for (int h = 0; h < 8; h++) {
        if (8 * q + h < n_draw) {
          const uint32_t word = (miso_block_half(hi, h) << 16) | miso_block_half(lo, h);
          const bool pick0 = miso_u01(word) * T < p0;
          d0 += pick0 ? 1 : 0;
          if (chain == 0) drawass[8 * q + h] = pick0 ? 0 : 1;
        }
      }
    }
    cnt0 = base0 + d0;
    cnt1 = base1 + (n_draw - d0);
  };
  auto mh_draws = [&](uint32_t iter, double &z, uint32_t &accept_word) {
    const miso_u32x4 b = miso_philox4x32(0u, iter, c2_mh, event_id, k0, k1);
    accept_word = b.v[0];
    z = miso_det_norm_from_unif(miso_u01(b.v[2]), miso_u01(b.v[3]));
  };
  auto propose = [&](double z, double &alphaN, double &x0, double &x1) {   // miso.c:449-471
    alphaN = alpha + c.sd * z;
    const double e = miso_det_exp(alphaN);
    const double sumexp = (0.0 + e) + 1.0;
    x0 = e / sumexp;
    x1 = 1 - (0.0 + x0);
  };

  {   // initial state (miso.c:362-369 K == 2: alpha = 0; miso.c:834, 841)
    double aN, x0, x1, z; uint32_t w;
    mh_draws(MISO_ITER_INIT, z, w);
    propose(z, aN, x0, x1);
    alpha = aN;
    cur = psi_terms(x0, x1, c.cst0, c.cst1);
  }
  // ILP: the exp / log coefficient tables of miso_detmath.h in VGPRs for the whole run (pinned: the empty asm keeps the
  // compiler from re-reading them from constant memory where they are used)
  double TE[12], TL[12];
  if constexpr (ILP) det_tables_to_registers(TE, TL);
  // the binomial's set-up for the chain's CURRENT psi (BinomSetup above); the proposal's is made beside its scores
  BinomSetup bs_cur{};
  if constexpr (ILP) {
    double sq_arg, lg_arg;
    binom_setup_pre(bs_cur, n_draw, cur.x0 / ((0.0 + cur.x1) + cur.x0), sq_arg, lg_arg);
    binom_setup_post(bs_cur, n_draw, det_sqrt_pos(sq_arg), miso_det_log(lg_arg), LF);
  }
  auto gibbs_fast = [&](uint32_t iter) {   // gibbs_collapsed with the set-up already made
    int d0 = 0;
    if (n_draw > 0) {
      d0 = binom_trials(bs_cur, n_draw, a.seed, event_id, chain, iter, LF, TL);
      if (!bs_cur.btrs) {   // few reads or psi near 0 / 1 (inversion, degenerate p): the routine as it stands
        miso_ustream us;
        miso_ustream_init(&us, a.seed, event_id, chain, iter, MISO_SITE_COUNTS);
        d0 = miso_binomial(&us, n_draw, bs_cur.p, a.logfact);
      }
    }
    cnt0 = base0 + d0;
    cnt1 = base1 + (n_draw - d0);
  };
  auto gibbs_step = [&](uint32_t iter) { if constexpr (ILP) gibbs_fast(iter); else gibbs_collapsed(iter); };
  if (a.M > 0) gibbs_step(MISO_ITER_INIT); else gibbs_per_read(MISO_ITER_INIT);

  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;
  RoundOpen ro(a);
  for (int m = 0; m < a.M; m++) {
    const bool opens = ro.at(a, m);   // a round's first iteration: no proposal terms (miso.c:866)
    hash = (hash ^ static_cast<uint32_t>(cnt0)) * 0x100000001B3ull;
    hash = (hash ^ static_cast<uint32_t>(cnt1)) * 0x100000001B3ull;
    if (trace) {
      int32_t *row = trace + (static_cast<size_t>(m) * a.C + chain) * 2;
      row[0] = cnt0; row[1] = cnt1;
    }
    double alphaN, x0, x1, z; uint32_t accept_word;
    mh_draws(static_cast<uint32_t>(m), z, accept_word);
    PsiTerms nw;
    double ptoCS, ctoPS;
    BinomSetup bs_new{};
    if constexpr (ILP) {
      // The step's eleven transcendentals in FIVE dependent stages (the values and the operations of psi_terms, propose,
      // prop_exponent above, in their order per value): exp {alpha', the proposal density's kernel around alpha'} -> log
      // {psi'_0, psi'_1, the logit, p(psi | alpha'), the binomial's log odds} beside the binomial's square root -> exp
      // {the two softmax terms, the other kernel} -> log {log-sum-exp, p(psi' | alpha)} -> exp {the acceptance ratio}.
      alphaN = alpha + c.sd * z;
      double e, xp;
      {
        const double in[2] = {alphaN, prop_exponent(cur.lgt, alphaN, c.sigma)};   // theta = psi, mu = alpha'
        double o[2];
        det_exp_n<2>(in, o, TE);
        e = o[0]; xp = o[1];
      }
      const double sumexp = (0.0 + e) + 1.0;
      x0 = e / sumexp;
      x1 = 1 - (0.0 + x0);
      double sq_arg, lg_arg;
      binom_setup_pre(bs_new, n_draw, x0 / ((0.0 + x1) + x0), sq_arg, lg_arg);
      nw.x0 = x0; nw.x1 = x1;
      const double ltheta = 1.0 - x0;
      double lpq;
      {
        const double in[5] = {x0, x1, x0 / ltheta, c.covar * cur.pr * xp, lg_arg};
        double o[5];
        det_log_n<5>(in, o, TL);
        nw.lx0 = o[0]; nw.lx1 = o[1]; nw.lgt = o[2]; ptoCS = o[3]; lpq = o[4];
      }
      binom_setup_post(bs_new, n_draw, det_sqrt_pos(sq_arg), lpq, LF);
      nw.pr = 1.0 / (1.0 * x0) / ltheta;
      const double lp0 = nw.lx0 + c.cst0, lp1 = nw.lx1 + c.cst1;
      const bool m1 = lp1 > lp0;  // miso.c:137-140: maxv starts at entry 0
      const double maxv = m1 ? lp1 : lp0;
      double ex0, ex1, xc;
      {
        const double in[3] = {lp0 - maxv, lp1 - maxv, prop_exponent(nw.lgt, alpha, c.sigma)};   // theta = psi', mu = alpha
        double o[3];
        det_exp_n<3>(in, o, TE);
        ex0 = o[0]; ex1 = o[1]; xc = o[2];
      }
      {
        const double in[2] = {(0.0 + ex0) + ex1, c.covar * nw.pr * xc};
        double o[2];
        det_log_n<2>(in, o, TL);
        const double lse = o[0] + maxv;
        nw.lpn0 = lp0 - lse;
        nw.lpn1 = lp1 - lse;
        ctoPS = o[1];
      }
    } else {
      propose(z, alphaN, x0, x1);
#ifdef MISO_LANE_NO_MH      // timing experiment: no scores
      nw = cur; nw.x0 = x0; nw.x1 = x1;
#else
      nw = psi_terms(x0, x1, c.cst0, c.cst1);
#endif
      const double xp = miso_det_exp(prop_exponent(cur.lgt, alphaN, c.sigma));   // theta = psi,  mu = alpha'
      const double xc = miso_det_exp(prop_exponent(nw.lgt, alpha, c.sigma));     // theta = psi', mu = alpha
      ptoCS = miso_det_log(c.covar * cur.pr * xp);
      ctoPS = miso_det_log(c.covar * nw.pr * xc);
    }
    const double pp = joint<false>(nw, cnt0, cnt1, c, 0.0);
    const double pc = joint<false>(cur, cnt0, cnt1, c, 0.0);
    double acceptP;
    if constexpr (ILP) {
      const double in[1] = {!opens ? pp + ptoCS - (pc + ctoPS) : pp - pc};
      double o[1];
      det_exp_n<1>(in, o, TE);
      acceptP = o[0];
    } else {
      acceptP = !opens ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);
    }
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    double cJS = pc;
    if (acc) { cur = nw; alpha = alphaN; cJS = pp; accepted++; if constexpr (ILP) bs_cur = bs_new; }
    if (m >= a.B) {  // miso.c:882-893
      if (lagCounter == a.lag - 1) {
        const size_t col = static_cast<size_t>(noS) + chain;
        *reinterpret_cast<double2 *>(samples + col * 2) = make_double2(cur.x0, cur.x1);
        loglik[col] = cJS;
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    if (m != a.M - 1) gibbs_step(static_cast<uint32_t>(m)); else gibbs_per_read(static_cast<uint32_t>(m));
  }
  hash = (hash ^ static_cast<uint32_t>(cnt0)) * 0x100000001B3ull;
  hash = (hash ^ static_cast<uint32_t>(cnt1)) * 0x100000001B3ull;
  if (trace) {
    int32_t *row = trace + (static_cast<size_t>(a.M) * a.C + chain) * 2;
    row[0] = cnt0; row[1] = cnt1;
  }
  ChainStats *st = reinterpret_cast<ChainStats *>(a.out_pool + E.off_stats) + chain;
  st->counts_hash = hash;
  st->accepted = accepted;
  st->hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
}

__global__ __launch_bounds__(256) void sampler_lane(const KernelArgs a) { lane_body<false>(a, nullptr); }
__global__ __launch_bounds__(256) void sampler_lane_ilp(const KernelArgs a) {
  __shared__ double lane_lf[LANE_LF_LDS];
  lane_body<MISO_LANE_ILP != 0>(a, lane_lf);
}

// The same step with G lanes per chain (k2_body COLLAPSED, kernels_k2.inl): a batch of fewer chains than the device has
// lanes -- 40 000 chains are 625 wavefronts of sampler_lane, one per SIMD on 60 % of the SIMDs, every dependent
// instruction exposed -- spreads each chain over 2 or 4 lanes: the Metropolis-Hastings step's transcendentals one per lane
// (vec_eval), the binomial's rejection trials G at a time.  Same results bit for bit.
#ifndef MISO_K2C_WAVES
#define MISO_K2C_WAVES 3
#endif
template <int G> __global__ __launch_bounds__(256, MISO_K2C_WAVES) void sampler_k2c(const KernelArgs a) {
  k2_body<G, 0, 4, false, true>(a, blockIdx.x, gridDim.x);
}
template __global__ void sampler_k2c<2>(const KernelArgs);
template __global__ void sampler_k2c<4>(const KernelArgs);
template __global__ void sampler_k2c<8>(const KernelArgs);

// ---- three and more isoforms (single-end): the same idea, per compatibility class ----
// A class of n reads compatible with isoforms v_0 < ... < v_{nv-1} gets counts Multinomial(n; psi_v / sum psi_v), drawn
// as a chain of binomials x_w ~ Binomial(n - x_0 - ... - x_{w-1}, psi_{v_w} / S_w), S_w = psi_{v_{nv-1}} + ... + psi_{v_w}
// summed in that order; classes in the order of the event's class table (= the counter order of the reads), one word
// stream per (chain, iteration).  One chain per lane; the chain's vectors live in LDS, [vector][isoform][lane], so
// that isoform loops need no unrolling and lanes never collide on a bank.  The Metropolis-Hastings step is the
// reference's arithmetic written out per chain (miso.c:97-163, 243-307, 449-552), terms of the current psi cached.
constexpr int LANEK_VECTORS = 9;   // alpha, alpha', psi, psi', log psi, log psi', log ratios, log ratios', scratch
// (LDS bytes per workgroup: LANEK_VECTORS x ks x 64 doubles + ks x 64 ints for the counts; runtime.hip lanek_lds_bytes)

__global__ __launch_bounds__(64) void sampler_lane_k(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_lane[];
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  const long slot = static_cast<long>(blockIdx.x) * 64 + threadIdx.x;
  if (slot >= n_chains) return;   // no barrier below
  const int lane = threadIdx.x, KS = a.kstride;
  double *vec = reinterpret_cast<double *>(smem_lane);
  int *cnt_base = reinterpret_cast<int *>(smem_lane + static_cast<size_t>(LANEK_VECTORS) * KS * 64 * 8);
#define LV(v, k) vec[(static_cast<size_t>(v) * KS + (k)) * 64 + lane]
#define CNT(k) cnt_base[(k) * 64 + lane]
  enum { ALPHA, ALPHAN, PSI, PSIN, LP, LPN, LR, LRN, TMP };
  const int ev = a.slot_event[slot / a.C];
  const uint32_t chain = static_cast<uint32_t>(slot % a.C);
  const DevEvent E = a.events[ev];
  const int K = E.K, len = K - 1;
  const uint32_t event_id = E.has_id ? E.explicit_id : a.first_event_id + static_cast<uint32_t>(ev);
  const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
  const double *cst = consts, *isc = consts + K, *hm1 = consts + 2 * K;
  const double lg_sum = consts[3 * K], lg_each = consts[3 * K + 1], sigma = consts[3 * K + 2], sd = consts[3 * K + 3],
               covar = consts[3 * K + 4];
  const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);
  const uint32_t *ctab = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_cls);
  const uint32_t *masks = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_draw);
  const int n_draw = E.n_draw, n_dcls = E.n_dcls;
  double *samples = reinterpret_cast<double *>(a.out_pool + E.off_samples);
  double *loglik = reinterpret_cast<double *>(a.out_pool + E.off_loglik);
  uint8_t *drawass = a.out_pool + E.off_drawass;
  int32_t *trace = (E.off_trace == NO_TRACE) ? nullptr : reinterpret_cast<int32_t *>(a.out_pool + E.off_trace);
  const uint32_t k0 = static_cast<uint32_t>(a.seed), k1 = static_cast<uint32_t>(a.seed >> 32);
  const uint32_t c2_gibbs = MISO_SITE_GIBBS | (chain << 8);

  // first drawing read of class c (device.hpp / host.cpp: row = {mask, units before, units before - first block, head | tail << 4})
  auto class_start = [&](int c) {
    if (c >= n_dcls) return n_draw;
    const uint32_t *row = ctab + CLS_WORDS * c;
    return static_cast<int>(4u * (row[1] - row[2]) + static_cast<uint32_t>(__builtin_ctz(row[3] & 0xFu)));
  };
  auto gibbs_collapsed = [&](uint32_t iter) {
    for (int k = 0; k < K; k++) CNT(k) = base[k];
    miso_ustream us;
    miso_ustream_init(&us, a.seed, event_id, chain, iter, MISO_SITE_COUNTS);
    int r0 = class_start(0);
    for (int c = 0; c < n_dcls; c++) {
      const uint32_t mask = ctab[CLS_WORDS * c];
      const int r1 = class_start(c + 1);
      int rem = r1 - r0;
      r0 = r1;
      double acc = 0.0;
      for (int k = K - 1; k >= 0; k--)
        if ((mask >> k) & 1u) { acc = acc + LV(PSI, k); LV(TMP, k) = acc; }
      const int last = 31 - __builtin_clz(mask);
      for (int k = 0; k < K; k++) {
        if (!((mask >> k) & 1u)) continue;
        const int x = (k == last) ? rem : miso_binomial(&us, rem, LV(PSI, k) / LV(TMP, k), a.logfact);
        CNT(k) += x;
        rem -= x;
      }
    }
  };
  // miso.c:30-91 read by read, word r of the Gibbs site for the r-th drawing read; chain 0's picks go back to the caller
  auto gibbs_per_read = [&](uint32_t iter) {
    for (int k = 0; k < K; k++) CNT(k) = base[k];
    miso_u32x4 u{};
    for (int r = 0; r < n_draw; r++) {
      if ((r & 3) == 0) u = miso_philox4x32(static_cast<uint32_t>(r >> 2), iter, c2_gibbs, event_id, k0, k1);
      const uint32_t word = (r & 3) == 0 ? u.v[0] : ((r & 3) == 1 ? u.v[1] : ((r & 3) == 2 ? u.v[2] : u.v[3]));
      const uint32_t mask = masks[r];
      const int nv = __builtin_popcount(mask);
      double total = 0.0;
      for (int k = 0; k < K; k++) if ((mask >> k) & 1u) total += LV(PSI, k);
      const double rnd = miso_u01(word) * total;
      const int lastk = 31 - __builtin_clz(mask);
      int sel = lastk;
      double acc = 0.0;
      for (int k = 0; k < K; k++) {
        if (!((mask >> k) & 1u)) continue;
        acc += LV(PSI, k);
        if (nv == 2) { if (rnd < acc) sel = k; break; }
        if (!(rnd > acc)) { sel = k; break; }
      }
      CNT(sel) += 1;
      if (chain == 0) drawass[r] = static_cast<uint8_t>(sel);
    }
  };
  // alpha' = alpha + sd z; psi' = logit_inv(alpha') (miso.c:184-241, 449-471)
  auto propose = [&](int from, int to_alpha, int to_psi, uint32_t iter, uint32_t &accept_word) {
    {
      const miso_u32x4 b0 = miso_draw_block(a.seed, event_id, chain, iter, MISO_SITE_MH, 0u);
      accept_word = b0.v[0];
    }
    double sumexp = 0.0;
    for (int i = 0; i < len; i++) {
      const int w = 2 + 2 * i;
      const miso_u32x4 b = miso_draw_block(a.seed, event_id, chain, iter, MISO_SITE_MH, static_cast<uint32_t>(w >> 2));
      const uint32_t w0 = (w & 3) == 0 ? b.v[0] : b.v[2], w1 = (w & 3) == 0 ? b.v[1] : b.v[3];
      const double z = miso_det_norm_from_unif(miso_u01(w0), miso_u01(w1));
      const double an = LV(from, i) + sd * z;
      LV(to_alpha, i) = an;
      const double e = miso_det_exp(an);
      LV(TMP, i) = e;
      sumexp += e;
    }
    sumexp += 1.0;
    double sumpsi = 0.0;
    for (int i = 0; i < len; i++) { const double x = LV(TMP, i) / sumexp; LV(to_psi, i) = x; sumpsi += x; }
    LV(to_psi, len) = 1 - sumpsi;
  };
  // what the two scores need of a psi and not of the counts: log psi_k, log(psi_k / (1 - sum)), 1 / prod / (1 - sum),
  // the log-sum-exp of log psi_k + cst_k (miso.c:104-113, 136-149)
  auto psi_cache = [&](int psi, int lp, int lr, double &jac, double &lse) {
    double ltheta = 1.0, prod = 1.0;
    for (int i = 0; i < len; i++) { const double t = LV(psi, i); ltheta -= t; prod *= t; }
    jac = 1.0 / prod / ltheta;
    for (int i = 0; i < len; i++) LV(lr, i) = miso_det_log(LV(psi, i) / ltheta);
    double maxv = 0.0;
    for (int i = 0; i < K; i++) {
      const double l = miso_det_log(LV(psi, i));
      LV(lp, i) = l;
      const double t = l + cst[i];
      maxv = (i == 0 || t > maxv) ? t : maxv;
    }
    double sum = 0.0;
    for (int i = 0; i < K; i++) sum += miso_det_exp((LV(lp, i) + cst[i]) - maxv);
    lse = miso_det_log(sum) + maxv;
  };
  auto joint = [&](int lp, double lse) {   // miso.c:243-307 from the counts
    double readProb = 0.0, assProb = 0.0, psiProb = 0.0;
    for (int i = 0; i < K; i++) {
      const int ci = CNT(i);
      if (ci != 0) {
        readProb = readProb + static_cast<double>(ci) * isc[i];
        assProb = assProb + static_cast<double>(ci) * ((LV(lp, i) + cst[i]) - lse);
      }
    }
    for (int i = 0; i < K; i++) psiProb += hm1[i] * LV(lp, i);
    psiProb += lg_sum;
    psiProb -= lg_each;
    return readProb + assProb + psiProb;
  };
  auto prop_score = [&](int lr, int mu, double jac) {   // miso.c:97-122
    double expPart = 0.0;
    for (int i = 0; i < len; i++) { const double t = LV(lr, i) - LV(mu, i); expPart += (-0.5) * t * t / sigma; }
    return miso_det_log(covar * jac * miso_det_exp(expPart));
  };

  // ---- initial state: miso.c:330-447, 834, 841 ----
  for (int i = 0; i < len; i++) LV(ALPHA, i) = (a.start == MISO_START_AUTO) ? 1.0 / (K - 1) : 0.0;
  uint32_t accept_word = 0;
  propose(ALPHA, ALPHA, PSI, MISO_ITER_INIT, accept_word);
  double jac = 0.0, lse = 0.0;
  psi_cache(PSI, LP, LR, jac, lse);
  if (a.M > 0) gibbs_collapsed(MISO_ITER_INIT); else gibbs_per_read(MISO_ITER_INIT);

  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;
  RoundOpen ro(a);
  for (int m = 0; m < a.M; m++) {
    const bool opens = ro.at(a, m);   // a round's first iteration: no proposal terms (miso.c:866)
    for (int i = 0; i < K; i++) {
      const int ci = CNT(i);
      hash = (hash ^ static_cast<uint32_t>(ci)) * 0x100000001B3ull;
      if (trace) trace[(static_cast<size_t>(m) * a.C + chain) * K + i] = ci;
    }
    propose(ALPHA, ALPHAN, PSIN, static_cast<uint32_t>(m), accept_word);
    double jacN, lseN;
    psi_cache(PSIN, LPN, LRN, jacN, lseN);
    const double pp = joint(LPN, lseN);
    const double pc = joint(LP, lse);
    const double ptoCS = prop_score(LR, ALPHAN, jac);      // theta = psi,  mu = alpha'
    const double ctoPS = prop_score(LRN, ALPHA, jacN);     // theta = psi', mu = alpha
    const double acceptP = !opens ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    double cJS = pc;
    if (acc) {
      for (int i = 0; i < K; i++) { LV(PSI, i) = LV(PSIN, i); LV(LP, i) = LV(LPN, i); }
      for (int i = 0; i < len; i++) { LV(ALPHA, i) = LV(ALPHAN, i); LV(LR, i) = LV(LRN, i); }
      jac = jacN; lse = lseN; cJS = pp; accepted++;
    }
    if (m >= a.B) {  // miso.c:882-893
      if (lagCounter == a.lag - 1) {
        const size_t col = static_cast<size_t>(noS) + chain;
        for (int i = 0; i < K; i++) samples[col * K + i] = LV(PSI, i);
        loglik[col] = cJS;
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    if (m != a.M - 1) gibbs_collapsed(static_cast<uint32_t>(m)); else gibbs_per_read(static_cast<uint32_t>(m));
  }
  for (int i = 0; i < K; i++) {
    const int ci = CNT(i);
    hash = (hash ^ static_cast<uint32_t>(ci)) * 0x100000001B3ull;
    if (trace) trace[(static_cast<size_t>(a.M) * a.C + chain) * K + i] = ci;
  }
  ChainStats *st = reinterpret_cast<ChainStats *>(a.out_pool + E.off_stats) + chain;
  st->counts_hash = hash;
  st->accepted = accepted;
  st->hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
#undef LV
#undef CNT
}

}  // namespace miso

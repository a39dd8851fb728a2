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

namespace miso {

__global__ __launch_bounds__(256) void sampler_lane(const KernelArgs a) {
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  const long slot = static_cast<long>(blockIdx.x) * 256 + threadIdx.x;
  if (slot >= n_chains) return;   // no barrier below
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
  // ... and read by read (miso.c:69-73: U (psi_0 + psi_1) < psi_0 picks isoform 0), word r of the Gibbs site for the
  // r-th drawing read; chain 0's picks are what the caller gets back
  auto gibbs_per_read = [&](uint32_t iter) {
    const double p0 = 0.0 + cur.x0, T = p0 + cur.x1;
    const int nq = (n_draw + 3) >> 2;
    int d0 = 0;
    for (int q = 0; q < nq; q++) {
      const miso_u32x4 u = miso_philox4x32_10(static_cast<uint32_t>(q), iter, c2_gibbs, event_id, k0, k1);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (4 * q + j < n_draw) {
          const bool pick0 = miso_u01(u.v[j]) * T < p0;
          d0 += pick0 ? 1 : 0;
          if (chain == 0) drawass[4 * q + j] = pick0 ? 0 : 1;
        }
      }
    }
    cnt0 = base0 + d0;
    cnt1 = base1 + (n_draw - d0);
  };
  auto mh_draws = [&](uint32_t iter, double &z, uint32_t &accept_word) {
    const miso_u32x4 b = miso_philox4x32_10(0u, iter, c2_mh, event_id, k0, k1);
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
  if (a.M > 0) gibbs_collapsed(MISO_ITER_INIT); else gibbs_per_read(MISO_ITER_INIT);

  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;
  for (int m = 0; m < a.M; m++) {
    hash = (hash ^ static_cast<uint32_t>(cnt0)) * 0x100000001B3ull;
    hash = (hash ^ static_cast<uint32_t>(cnt1)) * 0x100000001B3ull;
    if (trace) {
      int32_t *row = trace + (static_cast<size_t>(m) * a.C + chain) * 2;
      row[0] = cnt0; row[1] = cnt1;
    }
    double alphaN, x0, x1, z; uint32_t accept_word;
    mh_draws(static_cast<uint32_t>(m), z, accept_word);
    propose(z, alphaN, x0, x1);
#ifdef MISO_LANE_NO_MH      // timing experiment: no scores
    PsiTerms nw = cur; nw.x0 = x0; nw.x1 = x1;
#else
    const PsiTerms nw = psi_terms(x0, x1, c.cst0, c.cst1);
#endif
    const double xp = miso_det_exp(prop_exponent(cur.lgt, alphaN, c.sigma));   // theta = psi,  mu = alpha'
    const double xc = miso_det_exp(prop_exponent(nw.lgt, alpha, c.sigma));     // theta = psi', mu = alpha
    const double ptoCS = miso_det_log(c.covar * cur.pr * xp);
    const double ctoPS = miso_det_log(c.covar * nw.pr * xc);
    const double pp = joint<false>(nw, cnt0, cnt1, c, 0.0);
    const double pc = joint<false>(cur, cnt0, cnt1, c, 0.0);
    const double acceptP = (m > 0) ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    double cJS = pc;
    if (acc) { cur = nw; alpha = alphaN; cJS = pp; accepted++; }
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
    if (m != a.M - 1) gibbs_collapsed(static_cast<uint32_t>(m)); else gibbs_per_read(static_cast<uint32_t>(m));
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

}  // namespace miso

// kernels_marginal.hip -- sampler_marginal: algorithm = MISO_ALGO_MARGINAL and MISO_ALGO_CLASSES (single-end), one chain per LANE.
//
// The reference's second algorithm (splicing_miso with SPLICING_ALGO_MARGINAL: miso.c:272-283 inside the score,
// 800-808 for the match matrix, 841 / 895-898 for what it leaves out) keeps no assignment of reads to isoforms: the
// chain is Metropolis-Hastings on psi alone, and the reads enter through the marginal likelihood
//     sum over reads of log( sum_k match[k, read] / effective length_k * psi_k ),
// with the same Dirichlet prior, drift proposal and acceptance rule as the default algorithm (miso.c:97-241, 449-552).
// Reads compatible with the same isoforms contribute the same term, so the sum runs over the event's read CLASSES
// (device.hpp DevEvent::off_mcls: a weight per isoform and the number of reads; the order of the header's classes),
// count x log -- the counter contract's "sums over reads by counts", as for the default algorithm's scores.
// SPLICING_ALGO_CLASSES (miso.c:284-295, 788-803) is the same chain with other weights: the classes are the gene's
// POSSIBLE read classes (every start position of a read along every isoform, assignment.c:90-276), the weight of
// isoform k in class c the share of k's start positions that give c -- the table is made by the host (host.cpp
// attach_gene_classes) and this kernel does not know the difference.  An iteration is a few dozen
// transcendentals and no read loop at all: one chain per lane, the chain's vectors in LDS as [vector][isoform][lane]
// (sampler_lane_k's layout; 64 lanes per workgroup up to 32 isoforms, 32 lanes beyond: runtime.hip).
//
// The assignment the caller gets back (miso.c:936-946: one reassignment from the final psi, made when the run is over
// because this algorithm never made one) is drawn per read by chain 0, Gibbs words of (seed, event, chain 0,
// MISO_ITER_INIT).  Bit for bit against the CPU checker's counter mode (tests/test_gpu_marginal.py), whose stream mode
// is pinned to the real reference run with SPLICING_ALGO_MARGINAL.
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "miso_amd.h"
#include "miso_detmath.h"
#include "miso_philox.h"

namespace miso {

__global__ __launch_bounds__(64) void sampler_marginal(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_marginal[];
  const int L = static_cast<int>(blockDim.x);
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  const long slot = static_cast<long>(blockIdx.x) * L + threadIdx.x;
  if (slot >= n_chains) return;   // no barrier below
  const int lane = threadIdx.x, KS = a.kstride;
  double *vec = reinterpret_cast<double *>(smem_marginal);
#define LV(v, k) vec[(static_cast<size_t>(v) * KS + (k)) * L + lane]
  enum { ALPHA, ALPHAN, PSI, PSIN, LP, LPN, LR, LRN, TMP };
  static_assert(TMP + 1 == MARGINAL_VECTORS, "device.hpp MARGINAL_VECTORS");
  const int ev = a.slot_event[slot / a.C];
  const uint32_t chain = static_cast<uint32_t>(slot % a.C);
  const DevEvent E = a.events[ev];
  const int K = E.K, len = K - 1;
  const uint32_t event_id = E.has_id ? E.explicit_id : a.first_event_id + static_cast<uint32_t>(ev);
  const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
  const double *hm1 = consts + 2 * K;
  const double lg_sum = consts[3 * K], lg_each = consts[3 * K + 1], sigma = consts[3 * K + 2], sd = consts[3 * K + 3],
               covar = consts[3 * K + 4];
  const double *mcls = reinterpret_cast<const double *>(a.in_pool + E.off_mcls);   // K + 1 doubles per class (device.hpp)
  const int n_mcls = E.n_mcls;
  double *samples = reinterpret_cast<double *>(a.out_pool + E.off_samples);
  double *loglik = reinterpret_cast<double *>(a.out_pool + E.off_loglik);
  const uint32_t k0 = static_cast<uint32_t>(a.seed), k1 = static_cast<uint32_t>(a.seed >> 32);

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
  // what the scores need of a psi: log psi_k (the prior), log(psi_k / (1 - sum)) and 1 / prod / (1 - sum) (the proposal
  // density, miso.c:104-113)
  auto psi_cache = [&](int psi, int lp, int lr, double &jac) {
    double ltheta = 1.0, prod = 1.0;
    for (int i = 0; i < len; i++) { const double t = LV(psi, i); ltheta -= t; prod *= t; }
    jac = 1.0 / prod / ltheta;
    for (int i = 0; i < len; i++) LV(lr, i) = miso_det_log(LV(psi, i) / ltheta);
    for (int i = 0; i < K; i++) LV(lp, i) = miso_det_log(LV(psi, i));
  };
  // miso.c:243-307 for this algorithm: the marginal likelihood class by class (miso.c:272-283: a class whose sum is 0
  // adds nothing), no assignment term, the Dirichlet prior (miso.c:165-182)
  auto joint = [&](int psi, int lp) {
    double readProb = 0.0, psiProb = 0.0;
    for (int c = 0; c < n_mcls; c++) {
      const double *row = mcls + static_cast<size_t>(K + 1) * c;
      double s = 0.0;
      for (int k = 0; k < K; k++) s += row[k] * LV(psi, k);   // (the reference adds the other isoforms' zeros too)
      if (s != 0) readProb = readProb + row[K] * miso_det_log(s);
    }
    for (int i = 0; i < K; i++) psiProb += hm1[i] * LV(lp, i);
    psiProb += lg_sum;
    psiProb -= lg_each;
    return readProb + psiProb;
  };
  auto prop_score = [&](int lr, int mu, double jac) {   // miso.c:97-122
    double expPart = 0.0;
    for (int i = 0; i < len; i++) { const double t = LV(lr, i) - LV(mu, i); expPart += (-0.5) * t * t / sigma; }
    return miso_det_log(covar * jac * miso_det_exp(expPart));
  };

  // ---- initial state: miso.c:330-447, 834 ----
  for (int i = 0; i < len; i++) LV(ALPHA, i) = (a.start == MISO_START_AUTO && K != 2) ? 1.0 / (K - 1) : 0.0;
  uint32_t accept_word = 0;
  propose(ALPHA, ALPHA, PSI, MISO_ITER_INIT, accept_word);
  double jac = 0.0;
  psi_cache(PSI, LP, LR, jac);
  double pc = joint(PSI, LP);   // a function of psi alone: recomputed only when psi changes

  // (the parity instrumentation hashes every iteration's assignment counts: this algorithm has none, all zero)
  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;
  RoundOpen ro(a);
  for (int m = 0; m < a.M; m++) {
    const bool opens = ro.at(a, m);   // a round's first iteration: no proposal terms (miso.c:866)
    for (int i = 0; i < K; i++) hash = (hash ^ 0u) * 0x100000001B3ull;
    propose(ALPHA, ALPHAN, PSIN, static_cast<uint32_t>(m), accept_word);
    double jacN;
    psi_cache(PSIN, LPN, LRN, jacN);
    const double pp = joint(PSIN, LPN);
    const double ptoCS = prop_score(LR, ALPHAN, jac);      // theta = psi,  mu = alpha'
    const double ctoPS = prop_score(LRN, ALPHA, jacN);     // theta = psi', mu = alpha
    const double acceptP = !opens ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    if (acc) {
      for (int i = 0; i < K; i++) { LV(PSI, i) = LV(PSIN, i); LV(LP, i) = LV(LPN, i); }
      for (int i = 0; i < len; i++) { LV(ALPHA, i) = LV(ALPHAN, i); LV(LR, i) = LV(LRN, i); }
      jac = jacN; pc = pp; accepted++;
    }
    if (m >= a.B) {  // miso.c:882-893
      if (lagCounter == a.lag - 1) {
        const size_t col = static_cast<size_t>(noS) + chain;
        for (int i = 0; i < K; i++) samples[col * K + i] = LV(PSI, i);
        loglik[col] = pc;
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
  }
  for (int i = 0; i < K; i++) hash = (hash ^ 0u) * 0x100000001B3ull;

  // miso.c:936-946: the one reassignment, from the final psi; chain 0's picks go back to the caller (miso.c:11-91: among
  // the read's compatible isoforms with weights psi_k, word r of the Gibbs site for the r-th drawing read)
  if (chain == 0) {
    const int n_draw = E.n_draw;
    const uint32_t *lo = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_draw);
    const uint32_t *hi = lo + ((n_draw + 3) & ~3);
    uint8_t *drawass = a.out_pool + E.off_drawass;
    const uint32_t c2 = MISO_SITE_GIBBS;   // | chain 0 << 8
    miso_u32x4 u{};
    for (int r = 0; r < n_draw; r++) {
      uint32_t word;
      if (K == 2) {   // two isoforms: the uniform's two half-words (include/miso_philox.h, lazy low bits)
        word = miso_split_word(a.seed, event_id, 0u, MISO_ITER_INIT, static_cast<uint32_t>(r));
      } else {
        if ((r & 3) == 0) u = miso_philox4x32(static_cast<uint32_t>(r >> 2), MISO_ITER_INIT, c2, event_id, k0, k1);
        word = (r & 3) == 0 ? u.v[0] : ((r & 3) == 1 ? u.v[1] : ((r & 3) == 2 ? u.v[2] : u.v[3]));
      }
      const uint64_t mask = static_cast<uint64_t>(lo[r]) | (K > 32 ? static_cast<uint64_t>(hi[r]) << 32 : 0ull);
      const int nv = __builtin_popcountll(mask);
      double total = 0.0;
      for (int k = 0; k < K; k++) if ((mask >> k) & 1ull) total += LV(PSI, k);
      const double rnd = miso_u01(word) * total;
      int sel = 63 - __builtin_clzll(mask);
      double acc = 0.0;
      for (int k = 0; k < K; k++) {
        if (!((mask >> k) & 1ull)) continue;
        acc += LV(PSI, k);
        if (nv == 2) { if (rnd < acc) sel = k; break; }
        if (!(rnd > acc)) { sel = k; break; }
      }
      drawass[r] = static_cast<uint8_t>(sel);
    }
  }
  ChainStats *st = reinterpret_cast<ChainStats *>(a.out_pool + E.off_stats) + chain;
  st->counts_hash = hash;
  st->accepted = accepted;
  st->hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
#undef LV
}

}  // namespace miso

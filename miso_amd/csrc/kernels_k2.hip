// kernels_k2.hip -- the two-isoform single-end sampler (SE / RI / A3SS / A5SS / MXE events:
// BASELINE.json configs[1]), lane-packed for CDNA4.
//
// Why a second kernel: with K = 2 the per-iteration scalar math (propose + Metropolis-Hastings,
// miso.c:449-552: ~11 f64 transcendentals and ~13 f64 divisions, one long dependency chain) costs
// as much as the whole Gibbs sweep over ~500 ambiguous reads.  One wavefront per chain
// (sampler_wave) spends 63/64 of those issue slots on redundant lanes.  Here a chain owns only
// G lanes (G = 1..64, a power of two chosen by the host from the batch size):
//   * the scalar math runs once per G lanes -> 64/G chains share every f64 instruction;
//   * in the Gibbs step the G lanes stride over the chain's draw quads (one Philox4x32-10 block =
//     the uniforms of four consecutive ambiguous reads), counting picks of isoform 0 with ONE u32
//     compare per read: for two compatible isoforms the reference's test
//         U * (psi0 + psi1) < psi0        (miso.c:69-73)
//     is monotone in the 32-bit uniform, so it is replaced by  u < t  with the integer threshold
//     t = #{u : fl(fl(u 2^-32) (psi0+psi1)) < psi0}, found exactly once per iteration;
//   * a log2(G)-step cross-lane reduction gives the chain its count.
// Everything that only depends on the CURRENT psi (log psi, normalised log psi*eff, the
// proposal-density terms) is cached and swapped on acceptance: identical bits, half the
// transcendentals.  No LDS, no barriers; chain state lives in VGPRs.  Reads with fewer than two
// compatible isoforms never reach the device (host.hpp PackedEvent).
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "miso_amd.h"
#include "miso_detmath.h"
#include "miso_philox.h"

#pragma clang fp contract(off)

namespace miso {

namespace {

// everything the MH step needs about one psi = (x0, x1) that does not depend on the counts
struct PsiTerms {
  double x0, x1;      // psi
  double lx0, lx1;    // log psi_k                               (miso.c:136-138, 174)
  double lpn0, lpn1;  // log psi_k + cst_k - logsumexp           (miso.c:136-149)
  double lgt;         // log(psi_0 / (1 - psi_0))                (miso.c:113)
  double pr;          // 1 / psi_0 / (1 - psi_0)                 (miso.c:105-110)
};

__device__ __forceinline__ PsiTerms psi_terms(double x0, double x1, double cst0, double cst1) {
  PsiTerms t;
  t.x0 = x0; t.x1 = x1;
  t.lx0 = miso_det_log(x0);
  t.lx1 = miso_det_log(x1);
  const double lp0 = t.lx0 + cst0, lp1 = t.lx1 + cst1;
  const bool m1 = lp1 > lp0;  // miso.c:137-140: maxv starts at entry 0
  const double maxv = m1 ? lp1 : lp0;
  const double dmin = m1 ? lp0 - maxv : lp1 - maxv;
  const double dmax = m1 ? lp1 - maxv : lp0 - maxv;
  const double emin = miso_det_exp(dmin);
  double emax = 1.0;                       // det_exp(+-0) == 1 exactly
  if (!(dmax == 0.0)) emax = miso_det_exp(dmax);  // only when the maximum is not finite
  const double ex0 = m1 ? emin : emax, ex1 = m1 ? emax : emin;
  const double lse = miso_det_log((0.0 + ex0) + ex1) + maxv;
  t.lpn0 = lp0 - lse;
  t.lpn1 = lp1 - lse;
  const double ltheta = 1.0 - x0;
  t.lgt = miso_det_log(x0 / ltheta);
  t.pr = 1.0 / (1.0 * x0) / ltheta;
  return t;
}

struct K2Consts {
  double cst0, cst1, is0, is1, hm0, hm1, lg_sum, lg_each, sigma, sd, covar;
};

// miso.c:243-307 with the per-read sums taken from the counts
__device__ __forceinline__ double joint(const PsiTerms &t, int c0, int c1, const K2Consts &c) {
  double readProb = 0.0, assProb = 0.0, psiProb = 0.0;
  if (c0 != 0) { readProb = readProb + static_cast<double>(c0) * c.is0; assProb = assProb + static_cast<double>(c0) * t.lpn0; }
  if (c1 != 0) { readProb = readProb + static_cast<double>(c1) * c.is1; assProb = assProb + static_cast<double>(c1) * t.lpn1; }
  psiProb = psiProb + c.hm0 * t.lx0;
  psiProb = psiProb + c.hm1 * t.lx1;
  psiProb = psiProb + c.lg_sum;
  psiProb = psiProb - c.lg_each;
  return readProb + assProb + psiProb;
}

// miso.c:97-122 for len = 1: log density of theta under the logistic normal centred at mu
__device__ __forceinline__ double prop_score(const PsiTerms &t, double mu, const K2Consts &c) {
  const double tmp = t.lgt - mu;
  const double expPart = 0.0 + (-0.5) * tmp * tmp / c.sigma;
  const double pdf = c.covar * t.pr * miso_det_exp(expPart);
  return miso_det_log(pdf);
}

// #{u in [0, 2^32) : fl(fl(u * 2^-32) * T) < p0}  -- the reference's two-way draw as a threshold
__device__ __forceinline__ uint64_t k2_threshold(double p0, double T) {
  double est = p0 / T * 4294967296.0;
  if (!(est > 0.0)) est = 0.0;
  if (est > 4294967296.0) est = 4294967296.0;
  uint64_t t = static_cast<uint64_t>(est);
  for (int g = 0; g < 64 && t > 0 &&
                  !(static_cast<double>(static_cast<uint32_t>(t - 1)) * (1.0 / 4294967296.0) * T < p0); g++) t--;
  for (int g = 0; g < 64 && t < 4294967296ull &&
                  (static_cast<double>(static_cast<uint32_t>(t)) * (1.0 / 4294967296.0) * T < p0); g++) t++;
  return t;
}

}  // namespace

template <int G, int W>
__global__ __launch_bounds__(256, W) void sampler_k2(const KernelArgs a) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & (G - 1);
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  const long wave_first = (static_cast<long>(blockIdx.x) * 256 + (threadIdx.x & ~63)) / G;
  if (wave_first >= n_chains) return;  // whole wavefront idle
  long slot = (static_cast<long>(blockIdx.x) * 256 + threadIdx.x) / G;
  const bool live = slot < n_chains;   // dead groups shadow the last chain and store nothing
  if (!live) slot = n_chains - 1;

  const int ev = a.slot_event[slot / a.C];
  const uint32_t chain = static_cast<uint32_t>(slot % a.C);
  const DevEvent E = a.events[ev];
  const uint32_t event_id = a.first_event_id + static_cast<uint32_t>(ev);
  const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
  const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);
  K2Consts c;
  c.cst0 = consts[0]; c.cst1 = consts[1]; c.is0 = consts[2]; c.is1 = consts[3];
  c.hm0 = consts[4]; c.hm1 = consts[5]; c.lg_sum = consts[6]; c.lg_each = consts[7];
  c.sigma = consts[8]; c.sd = consts[9]; c.covar = consts[10];
  const int base0 = base[0], base1 = base[1];
  const int n_draw = E.n_draw, n_quads = (n_draw + 3) >> 2;
  int nq_wave = n_quads;  // loop bound must be wave-uniform: max over the wave's chains
  for (int off = 32; off >= 1; off >>= 1) nq_wave = max(nq_wave, __shfl_xor(nq_wave, off));

  double *samples = reinterpret_cast<double *>(a.out_pool + E.off_samples);
  double *loglik = reinterpret_cast<double *>(a.out_pool + E.off_loglik);
  uint32_t *drawass = reinterpret_cast<uint32_t *>(a.out_pool + E.off_drawass);
  int32_t *trace = (E.off_trace == NO_TRACE) ? nullptr
                                             : reinterpret_cast<int32_t *>(a.out_pool + E.off_trace);
  const uint32_t k0 = static_cast<uint32_t>(a.seed), k1 = static_cast<uint32_t>(a.seed >> 32);
  const uint32_t c2_gibbs = MISO_SITE_GIBBS | (chain << 8), c2_mh = MISO_SITE_MH | (chain << 8);

  int cnt0 = 0, cnt1 = 0;
  PsiTerms cur;
  double alpha = 0.0;

  auto gibbs = [&](uint32_t iter, bool write_ass) {
    const uint64_t t = k2_threshold(cur.x0, (0.0 + cur.x0) + cur.x1);
    const bool all = t >= 4294967296ull;
    const uint32_t t32 = static_cast<uint32_t>(t);
    int d0 = 0;
    for (int q = sub; q < nq_wave; q += G) {
      const miso_u32x4 u = miso_philox4x32_10(static_cast<uint32_t>(q), iter, c2_gibbs, event_id, k0, k1);
      if (q < n_quads) {
        const int rem = n_draw - 4 * q;  // >= 1 valid draws in this quad
        const int p0 = all | (u.v[0] < t32), p1 = all | (u.v[1] < t32), p2 = all | (u.v[2] < t32),
                  p3 = all | (u.v[3] < t32);
        d0 += p0 + (rem > 1 ? p1 : 0) + (rem > 2 ? p2 : 0) + (rem > 3 ? p3 : 0);
        if (write_ass)  // isoform index per read, one byte each: pick ? 0 : 1
          drawass[q] = (p0 ? 0u : 1u) | (p1 ? 0u : 1u) << 8 | (p2 ? 0u : 1u) << 16 | (p3 ? 0u : 1u) << 24;
      }
    }
#pragma unroll
    for (int off = G >> 1; off >= 1; off >>= 1) d0 += __shfl_xor(d0, off);
    cnt0 = base0 + d0;
    cnt1 = base1 + (n_draw - d0);
  };

  // alpha' = alpha + sd z, psi' = logit_inv(alpha') (miso.c:449-471); also hands back the accept word
  auto propose = [&](uint32_t iter, double &alphaN, double &x0, double &x1, uint32_t &accept_word) {
    const miso_u32x4 b = miso_philox4x32_10(0u, iter, c2_mh, event_id, k0, k1);
    accept_word = b.v[0];
    const double z = miso_det_norm_from_unif(miso_u01(b.v[2]), miso_u01(b.v[3]));
    alphaN = alpha + c.sd * z;
    const double e = miso_det_exp(alphaN);
    const double sumexp = (0.0 + e) + 1.0;
    x0 = e / sumexp;
    x1 = 1 - (0.0 + x0);
  };

  // ---- initial state (miso.c:362-369 K == 2: alpha = 0; miso.c:834, 841) ----
  {
    double aN, x0, x1; uint32_t w;
    propose(MISO_ITER_INIT, aN, x0, x1, w);
    alpha = aN;
    cur = psi_terms(x0, x1, c.cst0, c.cst1);
  }
  gibbs(MISO_ITER_INIT, live && sub < G && chain == 0 && a.M == 0);

  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;
  const bool writer = live && sub == 0;

  for (int m = 0; m < a.M; m++) {
    hash = (hash ^ static_cast<uint32_t>(cnt0)) * 0x100000001B3ull;
    hash = (hash ^ static_cast<uint32_t>(cnt1)) * 0x100000001B3ull;
    if (trace && writer) {
      int32_t *row = trace + (static_cast<size_t>(m) * a.C + chain) * 2;
      row[0] = cnt0; row[1] = cnt1;
    }
    double alphaN, x0, x1; uint32_t accept_word;
    propose(static_cast<uint32_t>(m), alphaN, x0, x1, accept_word);
    const PsiTerms nw = psi_terms(x0, x1, c.cst0, c.cst1);
    const double pp = joint(nw, cnt0, cnt1, c);
    const double pc = joint(cur, cnt0, cnt1, c);
    const double ptoCS = prop_score(cur, alphaN, c);
    const double ctoPS = prop_score(nw, alpha, c);
    const double acceptP = (m > 0) ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    double cJS = pc;
    if (acc) { cur = nw; alpha = alphaN; cJS = pp; accepted++; }

    if (m >= a.B) {  // miso.c:882-893
      if (lagCounter == a.lag - 1) {
        if (writer) {
          const size_t col = static_cast<size_t>(noS) + chain;
          *reinterpret_cast<double2 *>(samples + col * 2) = make_double2(cur.x0, cur.x1);
          loglik[col] = cJS;
        }
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    gibbs(static_cast<uint32_t>(m), live && chain == 0 && m == a.M - 1);
  }
  hash = (hash ^ static_cast<uint32_t>(cnt0)) * 0x100000001B3ull;
  hash = (hash ^ static_cast<uint32_t>(cnt1)) * 0x100000001B3ull;
  if (writer) {
    if (trace) {
      int32_t *row = trace + (static_cast<size_t>(a.M) * a.C + chain) * 2;
      row[0] = cnt0; row[1] = cnt1;
    }
    ChainStats *st = reinterpret_cast<ChainStats *>(a.out_pool + E.off_stats) + chain;
    st->counts_hash = hash;
    st->accepted = accepted;
    st->pad = 0;
  }
}

#define MISO_INSTANTIATE_K2(G) \
  template __global__ void sampler_k2<G, 2>(const KernelArgs); \
  template __global__ void sampler_k2<G, 3>(const KernelArgs); \
  template __global__ void sampler_k2<G, 4>(const KernelArgs);
MISO_INSTANTIATE_K2(1)
MISO_INSTANTIATE_K2(2)
MISO_INSTANTIATE_K2(4)
MISO_INSTANTIATE_K2(8)
MISO_INSTANTIATE_K2(16)
MISO_INSTANTIATE_K2(32)
MISO_INSTANTIATE_K2(64)

}  // namespace miso

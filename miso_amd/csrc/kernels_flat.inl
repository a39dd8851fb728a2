// kernels_flat.inl -- single-end sampler for three or more isoforms: a wavefront owns NC chains and
// every phase of an iteration is a flat loop over (chain, item) pairs on all 64 lanes.
// (template code; kernels_flat_c*.hip instantiate it per isoform-count class KC = 4, 8, 12, 16, 32)
//
// sampler_grp gives every chain a fixed group of G lanes.  Its profile (profiles/r01_grp_phase_cycles*)
// showed where that breaks down from ~5 isoforms on: the read loop's lanes stride over work units and
// hop over whole classes every trip (a chain of dependent LDS reads per hop), every (class, member)
// threshold re-sums the class's psi from scratch, and the LDS slice caps the chains per wavefront at
// four, so the per-wavefront cost of the scalar step is shared by four chains only.  Here:
//   * all per-chain state lives in an LDS slice (device.hpp FlatLayout); a wavefront carries as many
//     chains as fit 20 KB (6 at K = 10, 12 at K = 5, 21 at K = 3);
//   * Metropolis-Hastings (miso.c:449-552, 243-307) = the same six transcendental passes as
//     sampler_grp (qnorm -> exp -> log -> exp -> log -> exp), each a flat loop over (chain, argument):
//     lane utilisation no longer depends on how K relates to a lane group; the reference's
//     left-to-right sums run on one "leader" lane per chain, all chains of the wavefront in parallel;
//   * thresholds: one lane per (chain, class) walks the class's isoforms once for the total and once
//     for the running cumulative weight, turning each member's weight into the exact integer
//     threshold of the reference's test (miso.c:69-79) and the running maximum into the
//     isoform-indexed row the read loop compares with -- one pass, psi in registers;
//   * Gibbs (miso.c:30-91): the wavefront's work units (Philox block x class, host.hpp) of ALL its
//     chains form one list; lane l owns the contiguous range [l T, (l+1) T), so it walks classes and
//     chains in order (a class change is one row of LDS reads, no searching) and all 64 lanes are busy
//     until the list ends, whatever the chains' sizes.  Counts go to isoform-indexed register counters
//     D_k, flushed to the chain's slice when the lane moves to the next chain.
// Arithmetic, summation orders, tie rules and RNG addresses are those of sampler_grp / sampler_wave
// and of the CPU checker's counter mode (include/miso_philox.h): results are bit-identical.
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "miso_amd.h"
#include "miso_detmath.h"
#include "miso_philox.h"
#include "gibbs_rng.hpp"

#pragma clang fp contract(off)

#ifdef MISO_K2_PROFILE
#define FPROF_T(var) const uint64_t var = __builtin_readcyclecounter()
#define FPROF_ADD(acc, t0, t1) acc += (t1) - (t0)
#else
#define FPROF_T(var)
#define FPROF_ADD(acc, t0, t1)
#endif

namespace miso {

namespace {

// LDS traffic between lanes of ONE wavefront: program order is enough once the compiler may not move
// the accesses (no s_barrier: a chain never spans wavefronts)
__device__ __forceinline__ void fsync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// indices into a chain's double scalars (FlatLayout::sx) ...
enum { SX_SUMEXP = 0, SX_LTHETA, SX_MAXV, SX_E1, SX_E2, SX_X1, SX_X2, SX_LA0, SX_LA1, SX_LA2, SX_LR0, SX_LR1,
       SX_LR2, SX_SD, SX_SIGMA, SX_COVAR };
// ... and int scalars (FlatLayout::misc)
enum { MI_K = 0, MI_NDRAW, MI_NCLS, MI_NUNITS, MI_EVID, MI_CHAIN, MI_PAR, MI_ACCW, MI_C3K1, MI_P1LO, MI_P1HIK0,
       MI_EV, MI_SAMP_LO, MI_SAMP_HI, MI_TRACE_LO, MI_TRACE_HI, MI_USTART };

// The reference's draw compares rnd = fl(fl(u 2^-32) T) with a cumulative weight c: `rnd < c` when two
// isoforms are compatible, `!(rnd > c)` otherwise (miso.c:69-79).  Both are monotone in the 32-bit word
// u, so each is an integer threshold: the number of words for which the test holds (0 .. 2^32).  Same
// search as sampler_grp's draw_threshold, carried out on exactly representable doubles.
__device__ __forceinline__ bool fthr_pred(bool le, double u, double c, double T) {
  const double rnd = u * (1.0 / 4294967296.0) * T;
  const bool p = le ? !(rnd > c) : (rnd < c);
  return (u < 0.0) ? true : ((u >= 4294967296.0) ? false : p);
}
__device__ __forceinline__ double flat_threshold(bool le, double c, double T, double est) {
  est = (est > 0.0) ? est : 0.0;                      // also maps NaN to 0
  est = (est > 4294967296.0) ? 4294967296.0 : est;
  const double t0 = __builtin_floor(est);
  const int n = fthr_pred(le, t0 - 1.0, c, T) + fthr_pred(le, t0, c, T) + fthr_pred(le, t0 + 1.0, c, T);
  double t = t0 - 1.0 + static_cast<double>(n);
  if (!fthr_pred(le, t0 - 2.0, c, T) || fthr_pred(le, t0 + 2.0, c, T)) {  // exact fallback, rarely taken
    t = t0;
    for (int g = 0; g < 4096 && t > 0.0 && !fthr_pred(le, t - 1.0, c, T); g++) t = t - 1.0;
    for (int g = 0; g < 4096 && t < 4294967296.0 && fthr_pred(le, t, c, T); g++) t = t + 1.0;
  }
  return t < 0.0 ? 0.0 : t;
}

// The read loop of one wavefront-iteration for at most TW + 1 isoforms per chain.
// Lane state: chain s, class c, unit i (within the chain), the class's unit range and thresholds.
template <int TW>
__device__ __forceinline__ void flat_units(unsigned char *wbase, const FlatLayout &L, int tws, int ncw, int trips,
                                           int s0, int c0, int i0, int n_mine, uint32_t iter, uint32_t k0, uint32_t k1) {
  if (trips == 0) return;
  int s = s0, c = c0 - 1, i = i0;
  int uend = i0, ust = 0, qd = 0, ncls = 0;   // uend == i forces the first class load
  uint32_t hm = 0xFFu;
  uint32_t T[TW];
  int D[TW];
#pragma unroll
  for (int j = 0; j < TW; j++) { T[j] = 0u; D[j] = 0; }
  GibbsRng rng;
  rng.k0 = k0; rng.k1 = k1; rng.p1lo = 0; rng.p1hi = 0; rng.c3k1 = 0;
  uint32_t n0r0 = 0;
  bool fresh = true;   // chain constants not loaded yet
  for (int t = 0; t < trips; t++) {
    const bool active = t < n_mine;
    if (active && i == uend) {   // next class (of this chain or of the next one)
      const int *mi = reinterpret_cast<const int *>(wbase + s * L.bytes + L.misc);
      if (fresh) { ncls = mi[MI_NCLS]; }
      c++;
      if (!fresh && c == ncls) {   // next chain: hand the counters over first
        int *dl = reinterpret_cast<int *>(wbase + s * L.bytes + L.dl);
#pragma unroll
        for (int j = 0; j < TW; j++) { if (D[j]) atomicAdd(&dl[j], D[j]); D[j] = 0; }
        do { s++; mi = reinterpret_cast<const int *>(wbase + s * L.bytes + L.misc); } while (mi[MI_NUNITS] == 0);
        ncls = mi[MI_NCLS];
        c = 0; i = 0;
        fresh = true;
      }
      if (fresh) {
        rng.c3k1 = static_cast<uint32_t>(mi[MI_C3K1]); rng.p1lo = static_cast<uint32_t>(mi[MI_P1LO]);
        n0r0 = static_cast<uint32_t>(mi[MI_P1HIK0]) ^ iter;
        fresh = false;
      }
      const uint32_t *row = reinterpret_cast<const uint32_t *>(wbase + s * L.bytes + L.ctab) + CLS_WORDS * c;
      ust = static_cast<int>(row[1]); qd = static_cast<int>(row[2]); hm = row[3];
      uend = static_cast<int>(row[CLS_WORDS + 1]);
      const uint32_t *th = reinterpret_cast<const uint32_t *>(wbase + s * L.bytes + L.thr) + c * tws;
      const int tw = mi[MI_K] - 1;
#pragma unroll
      for (int j = 0; j < TW; j++) T[j] = (j < tw) ? th[j] : 0u;
    }
    uint32_t wm = active ? 0xFu : 0u;
    if (i == ust) wm &= hm;
    if (i == uend - 1) wm &= hm >> 4;
    const miso_u32x4 u = philox_gibbs<true>(rng, static_cast<uint32_t>(active ? i - qd : 0), n0r0);
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t uw = ((wm >> w) & 1u) ? u.v[w] : 0xFFFFFFFFu;   // never below a 32-bit threshold
#pragma unroll
      for (int j = 0; j < TW; j++) D[j] += (uw < T[j]) ? 1 : 0;
    }
    i += active ? 1 : 0;
  }
  if (n_mine > 0) {
    int *dl = reinterpret_cast<int *>(wbase + s * L.bytes + L.dl);
#pragma unroll
    for (int j = 0; j < TW; j++) if (D[j]) atomicAdd(&dl[j], D[j]);
  }
}

}  // namespace

template <int KC>
__global__ __launch_bounds__(256, 2) void sampler_flat(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_flat[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int NC = a.nc, ks = a.kstride, cs = a.cstride, tws = ks - 1;
  const FlatLayout L = flat_layout(ks, cs);
  unsigned char *wbase = smem_flat + static_cast<size_t>(wid) * NC * L.bytes;
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  const long wave_id = static_cast<long>(blockIdx.x) * 4 + wid;
  const long first_slot = wave_id * NC;
  if (first_slot >= n_chains) return;   // no block-level barrier below
  const int ncw = __builtin_amdgcn_readfirstlane(static_cast<int>(min(static_cast<long>(NC), n_chains - first_slot)));
  const uint32_t k0 = static_cast<uint32_t>(a.seed), k1 = static_cast<uint32_t>(a.seed >> 32);

#define FD(s, off) reinterpret_cast<double *>(wbase + (s) * L.bytes + (off))
#define FI(s, off) reinterpret_cast<int *>(wbase + (s) * L.bytes + (off))
#define FU(s, off) reinterpret_cast<uint32_t *>(wbase + (s) * L.bytes + (off))

  // ---- set-up: every chain's constants and class table into its slice ----
  int Kw = 0;
  for (int s = 0; s < ncw; s++) {
    const long slot = first_slot + s;
    const int ev = a.slot_event[slot / a.C];
    const uint32_t chain = static_cast<uint32_t>(slot % a.C);
    const DevEvent E = a.events[ev];
    const int K = E.K;
    Kw = max(Kw, K);
    const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
    const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);
    const uint32_t *gt = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_cls);
    const uint32_t event_id = E.has_id ? E.explicit_id : a.first_event_id + static_cast<uint32_t>(ev);
    for (int k = lane; k < K; k += 64) {
      FD(s, L.cst)[k] = consts[k]; FD(s, L.isc)[k] = consts[K + k]; FD(s, L.hm1)[k] = consts[2 * K + k];
      // miso.c:330-447 START_AUTO: K != 2 -> alpha = 1/(K-1); START_UNIFORM -> 0
      FD(s, L.alpha)[k] = (a.start == MISO_START_AUTO && K != 2 && k < K - 1) ? 1.0 / (K - 1) : 0.0;
      FD(s, L.alpha)[ks + k] = 0.0;
      FD(s, L.psi)[k] = 0.0; FD(s, L.psi)[ks + k] = 0.0;
      FI(s, L.cnt)[k] = 0; FI(s, L.bas)[k] = base[k]; FI(s, L.dl)[k] = 0;
      FU(s, L.ctab)[CLS_WORDS * (cs + 1) + k] = gt[CLS_WORDS * (E.n_dcls + 1) + k];   // A_k
    }
    for (int i = lane; i < CLS_WORDS * (E.n_dcls + 1); i += 64) FU(s, L.ctab)[i] = gt[i];
    if (lane == 0) {
      int *mi = FI(s, L.misc);
      mi[MI_K] = K; mi[MI_NDRAW] = E.n_draw; mi[MI_NCLS] = E.n_dcls; mi[MI_NUNITS] = E.n_units;
      mi[MI_EVID] = static_cast<int>(event_id); mi[MI_CHAIN] = static_cast<int>(chain);
      mi[MI_PAR] = 0; mi[MI_ACCW] = 0; mi[MI_EV] = ev;
      const GibbsRng g = gibbs_rng_init(a.seed, event_id, chain);
      mi[MI_C3K1] = static_cast<int>(g.c3k1); mi[MI_P1LO] = static_cast<int>(g.p1lo);
      mi[MI_P1HIK0] = static_cast<int>(g.p1hi ^ g.k0);
      const uint64_t so = E.off_samples, to = E.off_trace;
      mi[MI_SAMP_LO] = static_cast<int>(so); mi[MI_SAMP_HI] = static_cast<int>(so >> 32);
      mi[MI_TRACE_LO] = static_cast<int>(to); mi[MI_TRACE_HI] = static_cast<int>(to >> 32);
      double *sx = FD(s, L.sx);
      sx[SX_SIGMA] = consts[3 * K + 2]; sx[SX_SD] = consts[3 * K + 3]; sx[SX_COVAR] = consts[3 * K + 4];
    }
  }
  Kw = __builtin_amdgcn_readfirstlane(Kw);
  fsync();
  // the wavefront's unit list: chain s owns units [ustart_s, ustart_s + n_units_s)
  int total_units = 0;
  for (int s = 0; s < ncw; s++) {
    if (lane == 0) FI(s, L.misc)[MI_USTART] = total_units;
    total_units += FI(s, L.misc)[MI_NUNITS];
  }
  total_units = __builtin_amdgcn_readfirstlane(total_units);
  const int trips = (total_units + 63) / 64;   // units per lane
  fsync();
  // this lane's first unit: chain, class, unit within the chain (static for the whole run)
  int s0 = 0, c0 = 0, i0 = 0, n_mine = 0;
  {
    const int start = lane * trips;
    n_mine = max(0, min(trips, total_units - start));
    if (n_mine > 0) {
      for (int s = 0; s < ncw; s++) {
        const int us = FI(s, L.misc)[MI_USTART], nu = FI(s, L.misc)[MI_NUNITS];
        if (nu > 0 && us <= start) { s0 = s; i0 = start - us; }
      }
      const uint32_t *ct = FU(s0, L.ctab);
      const int ncls = FI(s0, L.misc)[MI_NCLS];
      for (int c = 0; c < ncls; c++) if (static_cast<int>(ct[CLS_WORDS * c + 1]) <= i0) c0 = c;
    }
  }

  // ---- the leader of chain s is lane s: the chain's sequential sums and its scalars ----
  const bool leader = lane < ncw;
  const int ls = leader ? lane : 0;
  const int lK = FI(ls, L.misc)[MI_K];
  const DevEvent LE_ = a.events[FI(ls, L.misc)[MI_EV]];
  const uint32_t lchain = static_cast<uint32_t>(FI(ls, L.misc)[MI_CHAIN]);
  double l_lg_sum = 0.0, l_lg_each = 0.0, l_covar = 0.0;
  {
    const double *consts = reinterpret_cast<const double *>(a.in_pool + LE_.off_consts);
    l_lg_sum = consts[3 * lK + 0]; l_lg_each = consts[3 * lK + 1]; l_covar = consts[3 * lK + 4];
  }
  double l_jac = 0.0, l_lse = 0.0;   // of the chain's current psi
  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;
  const float inv_k1 = 1.0f / static_cast<float>(tws), inv_k = 1.0f / static_cast<float>(ks),
              inv_2k = 1.0f / static_cast<float>(2 * ks - 1), inv_k2 = 1.0f / static_cast<float>(ks + 2),
              inv_cs = 1.0f / static_cast<float>(max(cs, 1));
  // flat loop over (chain s, item j < nper): idx = s * nper + j
#define FLAT_BEGIN(nper, inv)                                                   \
  for (int base_ = 0; base_ < ncw * (nper); base_ += 64) {                      \
    const int idx_ = base_ + lane;                                              \
    const bool on_ = idx_ < ncw * (nper);                                       \
    const int s = on_ ? static_cast<int>((static_cast<float>(idx_) + 0.5f) * (inv)) : 0; \
    const int j = on_ ? idx_ - s * (nper) : (nper);
#define FLAT_END }

#ifdef MISO_K2_PROFILE
  uint64_t fp_mh = 0, fp_thr = 0, fp_loop = 0;
#endif

  // ---- alpha' = alpha + sd z ; psi' = logit_inv(alpha') (miso.c:449-471), then the psi-only parts of
  // both scores of the new point: lp = log x, tb = lp + cst, lr = log(x_k / x_K'), jacobian.
  // sp / dp: source / destination buffer of every chain relative to its parity (0 = current). ----
  auto propose_and_logs = [&](uint32_t iter, int srel, int drel, double &jac_out) {
    FLAT_BEGIN(tws, inv_k1)   // pass 1 (qnorm) + pass 2 (exp), one normal per lane
      const int *mi = FI(s, L.misc);
      const int K = mi[MI_K];
      if (j < K - 1) {
        const int par = mi[MI_PAR];
        const int w = 2 + 2 * j;
        const miso_u32x4 b = miso_draw_block(a.seed, static_cast<uint32_t>(mi[MI_EVID]), static_cast<uint32_t>(mi[MI_CHAIN]),
                                             iter, MISO_SITE_MH, static_cast<uint32_t>(w >> 2));
        const bool odd = (j & 1) != 0;   // w & 3 = 0 for odd j, 2 for even j
        const double z = miso_det_norm_from_unif(miso_u01(odd ? b.v[0] : b.v[2]), miso_u01(odd ? b.v[1] : b.v[3]));
        const double an = FD(s, L.alpha)[(par ^ srel) * ks + j] + FD(s, L.sx)[SX_SD] * z;
        FD(s, L.alpha)[(par ^ drel) * ks + j] = an;
        FD(s, L.tc)[j] = miso_det_exp(an);
        if (j == 0) FI(s, L.misc)[MI_ACCW] = static_cast<int>(b.v[0]);   // block 0, word 0 (miso.c:870)
      }
    FLAT_END
    fsync();
    if (leader) {
      const double *tc = FD(ls, L.tc);
      double acc = 0.0;
      for (int k = 0; k < Kw - 1; k++) if (k < lK - 1) acc = acc + tc[k];
      FD(ls, L.sx)[SX_SUMEXP] = acc + 1.0;
    }
    fsync();
    FLAT_BEGIN(tws, inv_k1)
      const int *mi = FI(s, L.misc);
      if (j < mi[MI_K] - 1) FD(s, L.psi)[(mi[MI_PAR] ^ drel) * ks + j] = FD(s, L.tc)[j] / FD(s, L.sx)[SX_SUMEXP];
    FLAT_END
    fsync();
    if (leader) {
      double *x = FD(ls, L.psi) + (FI(ls, L.misc)[MI_PAR] ^ drel) * ks;
      double sumpsi = 0.0, ltheta = 1.0, prod = 1.0;
      for (int k = 0; k < Kw - 1; k++) if (k < lK - 1) { const double t = x[k]; sumpsi = sumpsi + t; ltheta = ltheta - t; prod = prod * t; }
      x[lK - 1] = 1 - sumpsi;
      FD(ls, L.sx)[SX_LTHETA] = ltheta;
      jac_out = 1.0 / prod / ltheta;
    }
    fsync();
    FLAT_BEGIN(2 * ks - 1, inv_2k)   // pass 3 (log): 2K - 1 arguments per chain
      const int *mi = FI(s, L.misc);
      const int K = mi[MI_K];
      const bool firsthalf = j < ks;
      const int k = firsthalf ? j : j - ks;
      if (firsthalf ? (k < K) : (k < K - 1)) {
        const int d = (mi[MI_PAR] ^ drel) * ks;
        const double xv = FD(s, L.psi)[d + k];
        const double r = miso_det_log(firsthalf ? xv : xv / FD(s, L.sx)[SX_LTHETA]);
        if (firsthalf) { FD(s, L.lp)[d + k] = r; FD(s, L.tb)[d + k] = r + FD(s, L.cst)[k]; }
        else FD(s, L.lr)[d + k] = r;
      }
    FLAT_END
    fsync();
  };
  auto leader_max = [&](int rel) {   // miso.c:137-140: maxv starts at entry 0
    double maxv = 0.0;
    if (leader) {
      const double *tb = FD(ls, L.tb) + (FI(ls, L.misc)[MI_PAR] ^ rel) * ks;
      maxv = tb[0];
      for (int k = 1; k < Kw; k++) if (k < lK) { const double v = tb[k]; if (v > maxv) maxv = v; }
      FD(ls, L.sx)[SX_MAXV] = maxv;
    }
    return maxv;
  };
  auto count_of = [&](int s, int k) { return FI(s, L.bas)[k] + FI(s, L.cnt)[k]; };
  // joint log score from cached logs and the current counts (miso.c:243-307), leader lanes
  auto joint_sums = [&](int rel, double lse) {
    const int d = (FI(ls, L.misc)[MI_PAR] ^ rel) * ks;
    const double *lp = FD(ls, L.lp) + d, *tb = FD(ls, L.tb) + d, *isc = FD(ls, L.isc), *hm1 = FD(ls, L.hm1);
    double readProb = 0.0, assProb = 0.0, psiProb = 0.0;
    for (int k = 0; k < Kw; k++) {
      if (k < lK) {
        const int ck = count_of(ls, k);
        if (ck != 0) {
          readProb = readProb + static_cast<double>(ck) * isc[k];
          assProb = assProb + static_cast<double>(ck) * (tb[k] - lse);
        }
      }
    }
    for (int k = 0; k < Kw; k++) if (k < lK) psiProb = psiProb + hm1[k] * lp[k];
    psiProb = psiProb + l_lg_sum;
    psiProb = psiProb - l_lg_each;
    return readProb + assProb + psiProb;
  };

  // ---- per-read picks by direct evaluation of the reference's scan (miso.c:11-22, 69-80): the final
  // assignment of chain 0 (miso.c:943-946) and the fallback when a threshold does not fit 32 bits ----
  auto direct_chain = [&](int s, uint32_t iter, bool count, bool write) {
    const int *mi = FI(s, L.misc);
    const int K = mi[MI_K], n_draw = mi[MI_NDRAW];
    const DevEvent E = a.events[mi[MI_EV]];
    const uint32_t *masks = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_draw);
    uint8_t *drawass = a.out_pool + E.off_drawass;
    const double *psi = FD(s, L.psi) + mi[MI_PAR] * ks;
    for (int q = lane; q < (n_draw + 3) / 4; q += 64) {
      const miso_u32x4 u = miso_draw_block(a.seed, static_cast<uint32_t>(mi[MI_EVID]), static_cast<uint32_t>(mi[MI_CHAIN]),
                                           iter, MISO_SITE_GIBBS, static_cast<uint32_t>(q));
      for (int w = 0; w < 4; w++) {
        const int r = 4 * q + w;
        if (r >= n_draw) break;
        const uint32_t m = masks[r];
        double T = 0.0; int nv = 0;
        for (int k = 0; k < K; k++) if ((m >> k) & 1u) { T = T + psi[k]; nv++; }
        const double rnd = miso_u01(u.v[w]) * T;
        double cum = 0.0; int idx = 0, sel = -1;
        for (int k = 0; k < K; k++) {
          if ((m >> k) & 1u) {
            cum = cum + psi[k];
            const bool stop = (nv == 2) ? (idx == 0 ? (rnd < cum) : true) : !(rnd > cum);
            idx++;
            if (sel < 0 && (stop || idx == nv)) sel = k;
          }
        }
        if (sel >= 0) {
          if (count) atomicAdd(&FI(s, L.cnt)[sel], 1);
          if (write) drawass[r] = static_cast<uint8_t>(sel);
        }
      }
    }
  };

  // ---- Gibbs step for every chain's current psi ----
  auto gibbs = [&](uint32_t iter) {
    FPROF_T(t0);
    // thresholds: one lane per (chain, class)
    bool slow = false;
    FLAT_BEGIN(cs, inv_cs)
      const int *mi = FI(s, L.misc);
      const int K = mi[MI_K];
      if (j < mi[MI_NCLS]) {
        const uint32_t m = FU(s, L.ctab)[CLS_WORDS * j];
        const double *psi = FD(s, L.psi) + mi[MI_PAR] * ks;
        double ps[KC];
#pragma unroll
        for (int k = 0; k < KC; k++) ps[k] = (k < K) ? psi[k] : 0.0;
        // total weight, ascending isoforms (miso.c:11-22); +0.0 for the others leaves the bits alone
        double T = 0.0;
#pragma unroll
        for (int k = 0; k < KC; k++) if (k < Kw) T = T + (((m >> k) & 1u) ? ps[k] : 0.0);
        const double inv = 4294967296.0 / T;
        const bool two = __popc(m) == 2;
        const int kmax = 31 - __clz(static_cast<int>(m));
        uint32_t *th = FU(s, L.thr) + j * tws;
        double cum = 0.0;
        uint32_t run = 0u;
#pragma unroll
        for (int k = 0; k < KC - 1; k++) {
          if (k < Kw - 1) {
            const bool member = (m >> k) & 1u;
            cum = cum + (member ? ps[k] : 0.0);
            uint32_t val = 0u;
            if (k < kmax) {   // a member before the last one: first j with u < t_j == u < max(t_0..t_j)
              if (member) {
                const double t = flat_threshold(!two, cum, T, cum * inv);
                slow |= t >= 4294967296.0;
                const uint32_t tu = static_cast<uint32_t>(t);
                run = tu > run ? tu : run;
              }
              val = run;
            }
            if (k < K - 1) th[k] = val;
          }
        }
      }
    FLAT_END
    FLAT_BEGIN(ks, inv_k)
      if (j < FI(s, L.misc)[MI_K]) { FI(s, L.dl)[j] = 0; FI(s, L.cnt)[j] = 0; }
    FLAT_END
    fsync();
    FPROF_T(t1);
    FPROF_ADD(fp_thr, t0, t1);
    if (__any(slow)) {   // a non-final threshold of 2^32 cannot be held in 32 bits: direct path this time
      for (int s = 0; s < ncw; s++) direct_chain(s, iter, true, false);
      fsync();
      return;
    }
    const int tww = Kw - 1;
#define MISO_FUNITS(TW) flat_units<TW>(wbase, L, tws, ncw, trips, s0, c0, i0, n_mine, iter, k0, k1);
    if constexpr (KC == 4) { if (tww <= 2) MISO_FUNITS(2) else MISO_FUNITS(3) }
    else if constexpr (KC == 8) { if (tww <= 4) MISO_FUNITS(4) else if (tww == 5) MISO_FUNITS(5) else if (tww == 6) MISO_FUNITS(6) else MISO_FUNITS(7) }
    else if constexpr (KC == 12) { if (tww <= 9) MISO_FUNITS(9) else MISO_FUNITS(11) }
    else if constexpr (KC == 16) { MISO_FUNITS(15) }
    else { if (tww <= 19) MISO_FUNITS(19) else if (tww <= 23) MISO_FUNITS(23) else MISO_FUNITS(31) }
#undef MISO_FUNITS
    fsync();
    // D_k (+ the reads of classes that end at or before k) -> picks per isoform
    FLAT_BEGIN(ks, inv_k)
      const int *mi = FI(s, L.misc);
      const int K = mi[MI_K];
      if (j < K) {
        const uint32_t *A = FU(s, L.ctab) + CLS_WORDS * (cs + 1);
        const int *dl = FI(s, L.dl);
        const int hi = (j < K - 1) ? dl[j] + static_cast<int>(A[j]) : mi[MI_NDRAW];
        const int lo = (j > 0) ? dl[j - 1] + static_cast<int>(A[j - 1]) : 0;
        FI(s, L.cnt)[j] = hi - lo;
      }
    FLAT_END
    fsync();
    FPROF_T(t2);
    FPROF_ADD(fp_loop, t1, t2);
  };

  // ---- initial state: miso.c:834 (alpha + sd z in place), cached logs, log-sum-exp, miso.c:841 ----
  propose_and_logs(MISO_ITER_INIT, 0, 0, l_jac);
  {
    const double maxv = leader_max(0);
    fsync();
    FLAT_BEGIN(ks, inv_k)
      const int *mi = FI(s, L.misc);
      if (j < mi[MI_K]) FD(s, L.tc)[j] = miso_det_exp(FD(s, L.tb)[mi[MI_PAR] * ks + j] - FD(s, L.sx)[SX_MAXV]);
    FLAT_END
    fsync();
    if (leader) {
      const double *tc = FD(ls, L.tc);
      double acc = 0.0;
      for (int k = 0; k < Kw; k++) if (k < lK) acc = acc + tc[k];
      l_lse = miso_det_log(acc) + maxv;
    }
    fsync();
  }
  gibbs(MISO_ITER_INIT);

  for (int m = 0; m < a.M; m++) {
    if (leader)
      for (int k = 0; k < Kw; k++)
        if (k < lK) hash = (hash ^ static_cast<uint32_t>(count_of(ls, k))) * 0x100000001B3ull;
    if (LE_.off_trace != NO_TRACE) {   // all events of a batch trace or none
      FLAT_BEGIN(ks, inv_k)
        const int *mi = FI(s, L.misc);
        const int K = mi[MI_K];
        if (j < K) {
          const uint64_t to = (static_cast<uint64_t>(static_cast<uint32_t>(mi[MI_TRACE_HI])) << 32) | static_cast<uint32_t>(mi[MI_TRACE_LO]);
          reinterpret_cast<int32_t *>(a.out_pool + to)[(static_cast<size_t>(m) * a.C + mi[MI_CHAIN]) * K + j] = count_of(s, j);
        }
      FLAT_END
    }
    FPROF_T(m0);
    double jacN = 0.0;
    propose_and_logs(static_cast<uint32_t>(m), 0, 1, jacN);                          // passes 1, 2, 3
    const double maxN = leader_max(1);
    // Gaussian parts of the two proposal densities (miso.c:110-117): proposal -> current uses the
    // current psi's log ratios against alpha', current -> proposal the proposal's against alpha
    FLAT_BEGIN(tws, inv_k1)
      const int *mi = FI(s, L.misc);
      if (j < mi[MI_K] - 1) {
        const int cu = mi[MI_PAR] * ks, pr = (mi[MI_PAR] ^ 1) * ks;
        const double sigma = FD(s, L.sx)[SX_SIGMA];
        const double t1 = FD(s, L.lr)[cu + j] - FD(s, L.alpha)[pr + j];
        FD(s, L.tc)[j] = (-0.5) * t1 * t1 / sigma;
        const double t2 = FD(s, L.lr)[pr + j] - FD(s, L.alpha)[cu + j];
        FD(s, L.u2)[j] = (-0.5) * t2 * t2 / sigma;
      }
    FLAT_END
    fsync();
    if (leader) {
      const double *tc = FD(ls, L.tc), *u2 = FD(ls, L.u2);
      double e1 = 0.0, e2 = 0.0;
      for (int k = 0; k < Kw - 1; k++) if (k < lK - 1) { e1 = e1 + tc[k]; e2 = e2 + u2[k]; }
      FD(ls, L.sx)[SX_E1] = e1; FD(ls, L.sx)[SX_E2] = e2;
    }
    fsync();
    FLAT_BEGIN(ks + 2, inv_k2)                                                         // pass 4: exp
      const int *mi = FI(s, L.misc);
      const int K = mi[MI_K];
      const bool iso = j < ks;
      if (iso ? (j < K) : true) {
        const double *sx = FD(s, L.sx);
        const double arg = iso ? FD(s, L.tb)[(mi[MI_PAR] ^ 1) * ks + j] - sx[SX_MAXV] : (j == ks ? sx[SX_E1] : sx[SX_E2]);
        const double r = miso_det_exp(arg);
        if (iso) FD(s, L.tc)[j] = r; else FD(s, L.sx)[SX_X1 + (j - ks)] = r;
      }
    FLAT_END
    fsync();
    if (leader) {
      const double *tc = FD(ls, L.tc);
      double *sx = FD(ls, L.sx);
      double sumtc = 0.0;
      for (int k = 0; k < Kw; k++) if (k < lK) sumtc = sumtc + tc[k];
      sx[SX_LA0] = sumtc; sx[SX_LA1] = l_covar * l_jac * sx[SX_X1]; sx[SX_LA2] = l_covar * jacN * sx[SX_X2];
    }
    fsync();
    for (int base_ = 0; base_ < ncw * 3; base_ += 64) {                                // pass 5: log
      const int idx_ = base_ + lane;
      if (idx_ < ncw * 3) {
        const int s = idx_ / 3, j = idx_ - 3 * s;
        FD(s, L.sx)[SX_LR0 + j] = miso_det_log(FD(s, L.sx)[SX_LA0 + j]);
      }
    }
    fsync();
    double cJS = 0.0;
    if (leader) {
      const double *sx = FD(ls, L.sx);
      const double lseN = sx[SX_LR0] + maxN, ptoCS = sx[SX_LR1], ctoPS = sx[SX_LR2];
      const double pp = joint_sums(1, lseN);
      const double pc = joint_sums(0, l_lse);
      const double acceptP = (m > 0) ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);  // pass 6
      const bool acc = (acceptP >= 1) || (miso_u01(static_cast<uint32_t>(FI(ls, L.misc)[MI_ACCW])) < acceptP);
      cJS = pc;
      if (acc) {   // the proposal and its cached logs become the current state
        FI(ls, L.misc)[MI_PAR] ^= 1;
        l_jac = jacN; l_lse = lseN;
        cJS = pp; accepted++;
      }
    }
    fsync();
    FPROF_T(m1);
    FPROF_ADD(fp_mh, m0, m1);
    if (m >= a.B) {  // miso.c:882-893
      if (lagCounter == a.lag - 1) {
        FLAT_BEGIN(ks, inv_k)
          const int *mi = FI(s, L.misc);
          const int K = mi[MI_K];
          if (j < K) {
            const uint64_t so = (static_cast<uint64_t>(static_cast<uint32_t>(mi[MI_SAMP_HI])) << 32) | static_cast<uint32_t>(mi[MI_SAMP_LO]);
            const size_t col = static_cast<size_t>(noS) + mi[MI_CHAIN];
            reinterpret_cast<double *>(a.out_pool + so)[col * K + j] = FD(s, L.psi)[mi[MI_PAR] * ks + j];
          }
        FLAT_END
        if (leader) reinterpret_cast<double *>(a.out_pool + LE_.off_loglik)[static_cast<size_t>(noS) + lchain] = cJS;
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    gibbs(static_cast<uint32_t>(m));
  }
  if (leader)
    for (int k = 0; k < Kw; k++)
      if (k < lK) hash = (hash ^ static_cast<uint32_t>(count_of(ls, k))) * 0x100000001B3ull;
  if (LE_.off_trace != NO_TRACE) {
    FLAT_BEGIN(ks, inv_k)
      const int *mi = FI(s, L.misc);
      const int K = mi[MI_K];
      if (j < K) {
        const uint64_t to = (static_cast<uint64_t>(static_cast<uint32_t>(mi[MI_TRACE_HI])) << 32) | static_cast<uint32_t>(mi[MI_TRACE_LO]);
        reinterpret_cast<int32_t *>(a.out_pool + to)[(static_cast<size_t>(a.M) * a.C + mi[MI_CHAIN]) * K + j] = count_of(s, j);
      }
    FLAT_END
  }
  // chain 0's final picks, read by read (miso.c:943-946): the last Gibbs step's draws once more
  for (int s = 0; s < ncw; s++)
    if (FI(s, L.misc)[MI_CHAIN] == 0)
      direct_chain(s, a.M > 0 ? static_cast<uint32_t>(a.M - 1) : MISO_ITER_INIT, false, true);
#ifdef MISO_K2_PROFILE
  if (leader && lchain == 0 && a.M > 8) {
    double *loglik = reinterpret_cast<double *>(a.out_pool + LE_.off_loglik);
    loglik[0] = static_cast<double>(fp_mh); loglik[1] = static_cast<double>(fp_thr); loglik[2] = static_cast<double>(fp_loop);
  }
#endif
  if (leader) {
    ChainStats *st = reinterpret_cast<ChainStats *>(a.out_pool + LE_.off_stats) + lchain;
    st->counts_hash = hash; st->accepted = accepted;
    st->hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#undef FD
#undef FI
#undef FU
#undef FLAT_BEGIN
#undef FLAT_END
}

}  // namespace miso

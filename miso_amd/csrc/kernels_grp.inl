// kernels_grp.inl -- lane-packed sampler for any isoform count and for paired-end reads.
// (template code; kernels_grp_c*.hip instantiate it per isoform-count class KC = 4, 8, 12, 16, 32:
// a kernel only carries the unrolled loop variants of its class, so the wide variants' register
// pressure does not leak into the narrow ones; runtime.hip launches each class's events separately)
//
// sampler_wave (kernels.hip) gives one wavefront to one chain; its per-iteration scalar step
// (miso.c:449-552: O(K) transcendentals in one dependency chain) keeps 64 lanes busy for the
// benefit of a single chain.  Here a chain owns G lanes (G = 2..32) and a wavefront carries 64/G
// chains, so that step is shared 64/G ways, exactly as in sampler_k2 -- but for any K <= 32 and for
// the paired-end model:
//   * per-chain vectors (psi, alpha, proposals, cached logs, per-isoform constants, counts) live in
//     an LDS slice of the chain; the reference's left-to-right sums are re-read from LDS by every
//     lane (broadcast reads), so all lanes of a chain hold identical scalars;
//   * Metropolis-Hastings step = SIX transcendental passes per iteration instead of one per call
//     site: every pass evaluates one function (qnorm, exp, log, exp, log, exp) on all the arguments
//     that are ready, one argument per lane -- the log psi'_k of the joint score next to the
//     log(psi'_k / psi'_K) of the proposal density, the softmax exponentials next to the two
//     Gaussian kernels, the log-sum-exp next to the two proposal densities.  Everything that
//     depends only on the CURRENT psi (its logs, log-sum-exp, Jacobian) is cached across
//     iterations and swapped in on acceptance.  Values and summation orders are those of
//     miso.c:97-163, 243-307, 449-552 (same as sampler_wave and the CPU checker);
//   * single-end Gibbs, class path: the drawing reads are ordered by compatibility class
//     (host.hpp), so a class's pick thresholds are integers computed once per iteration (one
//     (class, member) pair per lane) and a read costs one Philox word and K - 1 compare+add into
//     register counters indexed by ISOFORM (D_k = reads whose pick is isoform <= k), which need
//     no flushing when the class changes -- no per-read LDS traffic, no atomics (class_units);
//   * paired-end Gibbs: K u16 fragment indices per read into the fragment-probability table staged
//     in LDS; picks are counted with LDS atomics on the chain's slice; the fragment score is
//     accumulated in 2^-26 fixed point and reduced over the lanes.
// Same arithmetic, same order, same RNG addresses as sampler_wave and the CPU checker.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "device.hpp"
#include "miso_amd.h"
#include "miso_detmath.h"
#include "miso_philox.h"
#include "gibbs_rng.hpp"
#include "coop.hpp"

#pragma clang fp contract(off)

#ifdef MISO_K2_PROFILE
#define GPROF_T(var) const uint64_t var = __builtin_readcyclecounter()
#define GPROF_ADD(acc, t0, t1) acc += (t1) - (t0)
#else
#define GPROF_T(var)
#define GPROF_ADD(acc, t0, t1)
#endif

#ifndef MISO_PE_SDWA
#define MISO_PE_SDWA 1
#endif
#ifndef MISO_PE_ASM_TESTS
#define MISO_PE_ASM_TESTS 1
#endif
#ifndef MISO_PE_MAD
#define MISO_PE_MAD 1
#endif
#ifndef MISO_PE_AB_UPTO
#define MISO_PE_AB_UPTO 4   // pe_dense: two loop bodies with alternating record registers up to this many isoforms
#endif
#ifndef MISO_PE_MASKED_FROM
#define MISO_PE_MASKED_FROM 6   // paired-end quad loop: exec-masked form above this many isoforms, branch-free form up to it
#endif
#ifndef MISO_GRP_ILP_MAX_TW
#define MISO_GRP_ILP_MAX_TW 3   // single-end class path: two Philox blocks in flight per lane up to K = 4 (measured: +3% at K=3, a loss from K=5)
#endif

namespace miso {

namespace {

// LDS accesses of different lanes of one wavefront are ordered by program order once the compiler
// is told not to move them: a wavefront-scope fence is enough (no s_barrier: chains never span waves)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// The sequential sums, products and maxima of the Metropolis-Hastings step over a slice array, in CHUNKS of
// four: the chunk's values are read first (independent LDS reads, one round trip), then the chain runs in
// registers with the entries beyond n switched to the operation's neutral element (x + 0.0 on a sum that
// started from +0.0, x - 0.0, x * 1.0: all bit-neutral).  A loop over n reads the LDS once per step, each
// read a full round trip in front of a dependent add: a dozen such loops were most of the step's time from
// ten isoforms on (profiles/r02_pe_phase_experiments.txt).  nw = the wavefront's largest n (uniform bound).
// Chunk width MH_CH: the whole class at once up to eight isoforms paired-end (K=3 57.3k -> 63.2k events/s against
// chunks of four), four beyond (the registers of a wider chunk spill: K=10 19.9k -> 13.9k), one = the plain loop
// single-end (168-register budget: K=5 81k -> 30k with the class-wide form, 77.6k with chunks of four).
template <int MH_CH> __device__ __forceinline__ void load_chunk(const double *v, int k0, int n, double (&t)[MH_CH]) {
  const int last = max(n - 1, 0);
#pragma unroll
  for (int i = 0; i < MH_CH; i++) t[i] = v[min(k0 + i, last)];
}
template <int MH_CH, bool ONE = false> __device__ __forceinline__ double seq_sum_u(const double *v, int n, int nw) {
  double acc = 0.0;
#pragma unroll 1
  for (int k0 = 0; k0 < (ONE ? 1 : nw); k0 += MH_CH) {   // ONE: the chunk covers the class, a single pass
    double t[MH_CH];
    load_chunk<MH_CH>(v, k0, n, t);
#pragma unroll
    for (int i = 0; i < MH_CH; i++) acc = acc + (k0 + i < n ? t[i] : 0.0);
  }
  return acc;
}
template <int MH_CH, bool ONE = false> __device__ __forceinline__ double seq_max_u(const double *v, int n, int nw) {
  double m = v[0];
#pragma unroll 1
  for (int k0 = 0; k0 < (ONE ? 1 : nw); k0 += MH_CH) {
    double t[MH_CH];
    load_chunk<MH_CH>(v, k0, n, t);
#pragma unroll
    for (int i = 0; i < MH_CH; i++) m = (k0 + i > 0 && k0 + i < n && t[i] > m) ? t[i] : m;
  }
  return m;
}

__device__ __forceinline__ double seq_sum(const double *v, int n) {
  double acc = 0.0;
  for (int k = 0; k < n; k++) acc = acc + v[k];
  return acc;
}

struct Slice {            // one chain's LDS slice; every per-isoform array has `ks` entries
  double *psi, *alpha, *lp, *tb, *lr;        // current state and its cache: log psi, log psi + cst, log(psi_k / psi_K')
  double *psiN, *alphaN, *lpN, *tbN, *lrN;   // the proposal's
  double *tc, *u2;                           // scratch (tc doubles as the first Gaussian-term buffer)
  double *cst, *isc, *hm1;                   // per-isoform constants (device.hpp)
  double *sx;                                // 4 scalars
  int *cnt, *bas;  // picks of the drawing reads; reads with a single compatible isoform
  int *dl;         // SE class path: D_k = drawing reads that picked an isoform <= k
  // SE class path (only when cstride > 0), per drawing-read class c:
  uint32_t *thr;   //   thr[c][k] = words below it pick an isoform <= k [c x (ks - 1)]
  uint32_t *ctab;  //   CLS_WORDS words per class + one sentinel row, then A_k (host.hpp dcls_tab)
  uint16_t *pairs; //   (class << 8 | isoform) of every threshold to compute
  int32_t *stab;   // PE: the event's fixed-point score table [tstride] (when it fits)
};

__device__ __forceinline__ Slice carve(unsigned char *base, int ks, int cs) {
  Slice s;
  double *d = reinterpret_cast<double *>(base);
  s.psi = d; s.alpha = d + ks; s.lp = d + 2 * ks; s.tb = d + 3 * ks; s.lr = d + 4 * ks;
  s.psiN = d + 5 * ks; s.alphaN = d + 6 * ks; s.lpN = d + 7 * ks; s.tbN = d + 8 * ks; s.lrN = d + 9 * ks;
  s.tc = d + 10 * ks; s.u2 = d + 11 * ks;
  s.cst = d + 12 * ks; s.isc = d + 13 * ks; s.hm1 = d + 14 * ks;
  s.sx = d + 15 * ks;
  s.cnt = reinterpret_cast<int *>(s.sx + 4);
  s.bas = s.cnt + ks;
  s.dl = s.bas + ks;
  unsigned char *c = reinterpret_cast<unsigned char *>(s.dl + ks + (ks & 1));
  const int tw = ks - 1;
  s.thr = reinterpret_cast<uint32_t *>(c);
  s.ctab = s.thr + cs * tw + ((cs * tw) & 1);
  s.pairs = reinterpret_cast<uint16_t *>(s.ctab + CLS_WORDS * (cs + 1) + ks + (ks & 1));
  s.stab = reinterpret_cast<int32_t *>(c + grp_cls_bytes(ks, cs));
  return s;
}

struct Scalars { double lg_sum, lg_each, sigma, sd, covar; };

// One read's pick by direct evaluation of the reference's scan (miso.c:11-22, 69-80; paired-end
// miso_paired.c:11-22, 64-75) for at most KK isoforms, fully unrolled, everything in registers.
template <int KK, bool PE>
__device__ __forceinline__ int pick_direct(const double (&ps)[8], uint32_t mask, const uint16_t *fr_row,
                                           int K, const double *lds_fp, bool on, uint32_t uword,
                                           uint16_t &fsel) {
  double w[KK]; uint16_t fr[KK]; bool val[KK];
  double T = 0.0; int nv = 0;
#pragma unroll
  for (int k = 0; k < KK; k++) {
    fr[k] = FRAG_NONE;
    if (PE) {
      if (on && k < K) fr[k] = fr_row[k];
      val[k] = fr[k] != FRAG_NONE;
      w[k] = ps[k] * lds_fp[val[k] ? fr[k] : 0];
    } else {
      val[k] = on && ((mask >> k) & 1u);
      w[k] = ps[k];
    }
    if (val[k]) { T = T + w[k]; nv++; }
  }
  const double rnd = miso_u01(uword) * T;
  double cum = 0.0; int idx = 0, sel = -1;
#pragma unroll
  for (int k = 0; k < KK; k++) {
    if (val[k]) {
      cum = cum + w[k];
      const bool stop = (nv == 2) ? (idx == 0 ? (rnd < cum) : true) : !(rnd > cum);
      idx++;
      if (sel < 0 && (stop || idx == nv)) { sel = k; fsel = fr[k]; }
    }
  }
  return sel;
}

// The reference's draw compares rnd = fl(fl(u 2^-32) T) with a cumulative weight c: `rnd < c` when two
// isoforms are compatible, `!(rnd > c)` otherwise (miso.c:69-79).  Both are monotone in the 32-bit
// word u, so each becomes an integer threshold: the number of words for which the test holds.
template <bool LE> __device__ __forceinline__ bool thr_pred(int64_t u, double c, double T) {
  if (u < 0) return true;
  if (u >= 4294967296ll) return false;
  const double rnd = static_cast<double>(static_cast<uint32_t>(u)) * (1.0 / 4294967296.0) * T;
  return LE ? !(rnd > c) : (rnd < c);
}
// `est` only seeds the search (any value gives the exact threshold; a good one makes it short)
template <bool LE> __device__ __forceinline__ uint64_t draw_threshold(double c, double T, double est) {
  est = (est > 0.0) ? est : 0.0;                      // also maps NaN to 0
  est = (est > 4294967296.0) ? 4294967296.0 : est;
  const int64_t t0 = static_cast<int64_t>(est);
  const int n = thr_pred<LE>(t0 - 1, c, T) + thr_pred<LE>(t0, c, T) + thr_pred<LE>(t0 + 1, c, T);
  int64_t t = t0 - 1 + n;
  if (!thr_pred<LE>(t0 - 2, c, T) || thr_pred<LE>(t0 + 2, c, T)) {  // exact fallback, rarely taken
    t = t0;
    for (int g = 0; g < 4096 && t > 0 && !thr_pred<LE>(t - 1, c, T); g++) t--;
    for (int g = 0; g < 4096 && t < 4294967296ll && thr_pred<LE>(t, c, T); g++) t++;
  }
  return static_cast<uint64_t>(t < 0 ? 0 : t);
}

// The single-end class path's read loop.  The chain's G lanes stride over its work units (unit i =
// the words of one Philox block that belong to one class; classes are consecutive, so units are
// implicit: class c owns units [ustart_c, ustart_(c+1)), unit i covers block i - qd_c, all four words
// except in the class's first / last unit).  A lane keeps the thresholds T[k] of its current class
// in registers and counts D[k] += (word < T[k]): isoform-indexed, so nothing is flushed when the
// class changes; the next class's row is prefetched while the current one is being consumed.
// TW = the wavefront's K - 1, rounded up to an instantiated value.
// One strided walk over a chain's work units: the class the current unit belongs to, that class's
// thresholds in registers, and the next class's row prefetched.
template <int TW> struct UnitWalker {
  uint32_t T[TW], NT[TW];
  int cur, ust, uend, qd;
  uint32_t hmtm;
  int n_ust, n_uend, n_qd;
  uint32_t n_hm;

  __device__ __forceinline__ void init(const uint32_t *ctab, const uint32_t *thr, int tw) {
#pragma unroll
    for (int j = 0; j < TW; j++) { T[j] = 0; NT[j] = (j < tw) ? thr[j] : 0u; }
    cur = -1; ust = 0; uend = 0; qd = 0; hmtm = 0xFFu;
    n_ust = static_cast<int>(ctab[1]); n_uend = static_cast<int>(ctab[CLS_WORDS + 1]);
    n_qd = static_cast<int>(ctab[2]); n_hm = ctab[3];
  }
  // make unit i the current one (i only grows); returns the word mask of the unit
  __device__ __forceinline__ uint32_t seek(const uint32_t *ctab, const uint32_t *thr, int tw, int ncls,
                                           int i, bool active) {
    if (active && i >= uend) {
      cur++;
      if (i >= n_uend) {   // the stride jumped over whole classes (classes of fewer units than the stride)
        do { cur++; n_uend = static_cast<int>(ctab[CLS_WORDS * (cur + 1) + 1]); } while (i >= n_uend);
        n_ust = static_cast<int>(ctab[CLS_WORDS * cur + 1]);
        n_qd = static_cast<int>(ctab[CLS_WORDS * cur + 2]);
        n_hm = ctab[CLS_WORDS * cur + 3];
#pragma unroll
        for (int j = 0; j < TW; j++) NT[j] = (j < tw) ? thr[cur * tw + j] : 0u;
      }
      ust = n_ust; uend = n_uend; qd = n_qd; hmtm = n_hm;
#pragma unroll
      for (int j = 0; j < TW; j++) T[j] = NT[j];
      const int nx = min(cur + 1, ncls - 1);
      n_ust = static_cast<int>(ctab[CLS_WORDS * nx + 1]);
      n_uend = static_cast<int>(ctab[CLS_WORDS * (nx + 1) + 1]);
      n_qd = static_cast<int>(ctab[CLS_WORDS * nx + 2]);
      n_hm = ctab[CLS_WORDS * nx + 3];
#pragma unroll
      for (int j = 0; j < TW; j++) NT[j] = (j < tw) ? thr[nx * tw + j] : 0u;
    }
    uint32_t wm = active ? 0xFu : 0u;
    if (i == ust) wm &= hmtm;
    if (i == uend - 1) wm &= hmtm >> 4;
    return wm;
  }
  __device__ __forceinline__ void count(const miso_u32x4 &u, uint32_t wm, int (&D)[TW]) const {
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t uw = ((wm >> w) & 1u) ? u.v[w] : 0xFFFFFFFFu;   // never below a 32-bit threshold
#pragma unroll
      for (int j = 0; j < TW; j++) D[j] += (uw < T[j]) ? 1 : 0;
    }
  }
};

// NW walkers per lane (units sub + G (NW m + w), m = 0, 1, ..): NW independent Philox blocks are in
// flight per trip.  One block's nine dependent rounds leave the multiplier idle between rounds and a
// batch of 40 000 chains only gives ~2.4 wavefronts per SIMD to fill the gaps, so for small K (few
// compares per word) a second block per lane is what hides the latency; for large K the TW
// compare-adds per word already do and the second walker's registers are not worth it.
template <int TW, int G, int NW>
__device__ __forceinline__ void class_units_nw(const uint32_t *ctab, const uint32_t *thr, int tw, int ncls,
                                               int nuw, int n_units, int sub, const GibbsRng &rng,
                                               uint32_t n0r0, int (&D)[TW]) {
  UnitWalker<TW> W[NW];
#pragma unroll
  for (int w = 0; w < NW; w++) W[w].init(ctab, thr, tw);
#pragma unroll
  for (int j = 0; j < TW; j++) D[j] = 0;
  for (int i0 = 0; i0 < nuw; i0 += NW * G) {
    uint32_t wm[NW]; miso_u32x4 u[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const int i = i0 + w * G + sub;
      const bool active = i < n_units;
      wm[w] = W[w].seek(ctab, thr, tw, ncls, i, active);
      u[w] = philox_gibbs<true>(rng, static_cast<uint32_t>(active ? i - W[w].qd : 0), n0r0);
    }
#pragma unroll
    for (int w = 0; w < NW; w++) W[w].count(u[w], wm[w], D);
  }
}

template <int TW, int G>
__device__ __forceinline__ void class_units(const uint32_t *ctab, const uint32_t *thr, int tw, int ncls,
                                            int nuw, int n_units, int sub, const GibbsRng &rng,
                                            uint32_t n0r0, int (&D)[TW]) {
  class_units_nw<TW, G, (TW <= MISO_GRP_ILP_MAX_TW) ? 2 : 1>(ctab, thr, tw, ncls, nuw, n_units, sub, rng, n0r0, D);
}

// The paired-end read loop when every chain of the wavefront has exactly KK isoforms.  A lane takes
// the quad of reads 4q .. 4q+3: their 4 KK fragment indices are one contiguous block of 2 KK dwords,
// fetched with wide loads ONE TRIP AHEAD so the L2 latency overlaps the previous quad's arithmetic;
// weights psi_k fp[f] from the LDS tables, pick by the reference's scan (miso_paired.c:11-22,
// 64-75), score from the chain's LDS table.  Reads beyond n_draw are padded with FRAG_NONE (host).
// Exec-masked formulation (data-dependent branches skip the isoforms a read is not compatible with):
// slower than the branch-free one below up to ~6 isoforms (K=3 +4%, K=5 +5%), faster beyond
// (K=10: 9.7k vs 7.1k events/s) where the selects of the branch-free form outnumber the work saved.
template <int KK, int G>
__device__ __forceinline__ void pe_quads_masked(const uint16_t *frags, const double *psi, const double *lds_fp,
                                         const int32_t *stab, int il, int *cnt, uint8_t *drawass,
                                         bool write_ass, int nqw, int n_quads, int n_draw, int sub,
                                         const GibbsRng &rng, uint32_t n0r0, int64_t &acc_out, int &bad_out) {
  constexpr int ND = 2 * KK;   // dwords per quad
  double ps[KK];
#pragma unroll
  for (int k = 0; k < KK; k++) ps[k] = psi[k];
  const uint32_t *fq = reinterpret_cast<const uint32_t *>(frags);
  uint32_t nxt[ND];
  {
    const int q = sub;
#pragma unroll
    for (int i = 0; i < ND; i++) nxt[i] = (q < n_quads) ? fq[static_cast<size_t>(q) * ND + i] : 0xFFFFFFFFu;
  }
  int64_t acc = 0; int bad = 0;
  for (int q0 = 0; q0 < nqw; q0 += G) {
    const int q = q0 + sub;
    uint32_t cur[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) cur[i] = nxt[i];
    {
      const int qn = q + G;
#pragma unroll
      for (int i = 0; i < ND; i++) nxt[i] = (qn < n_quads) ? fq[static_cast<size_t>(qn) * ND + i] : 0xFFFFFFFFu;
    }
    const miso_u32x4 u = philox_gibbs<true>(rng, static_cast<uint32_t>(q), n0r0);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t fr[KK]; double w[KK]; bool val[KK];
      double T = 0.0; int nv = 0;
#pragma unroll
      for (int k = 0; k < KK; k++) {
        const int idx = j * KK + k;
        fr[k] = (cur[idx >> 1] >> (16 * (idx & 1))) & 0xFFFFu;
        val[k] = fr[k] != FRAG_NONE;
        w[k] = ps[k] * lds_fp[val[k] ? fr[k] : 0];
        if (val[k]) { T = T + w[k]; nv++; }
      }
      const double rnd = miso_u01(u.v[j]) * T;
      double cum = 0.0; int idx = 0, sel = -1; uint32_t fsel = 0;
#pragma unroll
      for (int k = 0; k < KK; k++) {
        if (val[k]) {
          cum = cum + w[k];
          const bool stop = (nv == 2) ? (idx == 0 ? (rnd < cum) : true) : !(rnd > cum);
          idx++;
          if (sel < 0 && (stop || idx == nv)) { sel = k; fsel = fr[k]; }
        }
      }
      if (sel >= 0) {
        atomicAdd(&cnt[sel], 1);   // with many isoforms one LDS atomic beats KK register updates
        const int32_t v = stab[sel * il + static_cast<int>(fsel)];
        if (v == SFIX_BAD) bad = 1; else acc += v;
        if (write_ass) drawass[4 * q + j] = static_cast<uint8_t>(sel);
      }
    }
  }
  acc_out = acc; bad_out = bad;
}


template <int KK, int G, bool WRITE>
__device__ __forceinline__ void pe_quads(const uint16_t *frags, const double *psi, const double *lds_fp,
                                         const int32_t *stab, int il, int *cnt, uint8_t *drawass,
                                         bool write_ass, int nqw, int n_quads, int n_draw, int sub,
                                         const GibbsRng &rng, uint32_t n0r0, int64_t &acc_out, int &bad_out) {
  // Written without data-dependent branches: the first version had ~10 per read (guarded loads,
  // `if (valid)` in the scan), each a scheduling barrier with its own s_waitcnt, and the loop ran at
  // LDS / L2 latency.  Invalid isoforms contribute +0.0 to the sums (bit-neutral: every weight is
  // >= +0), out-of-range quads are loaded from a clamped address and masked.
  constexpr int ND = 2 * KK;   // dwords per quad
  double ps[KK];
#pragma unroll
  for (int k = 0; k < KK; k++) ps[k] = psi[k];
  const uint32_t *fq = reinterpret_cast<const uint32_t *>(frags);
  const int q_last = max(n_quads - 1, 0);
  uint32_t nxt[ND];
  {
    const int q = sub;
    const uint32_t *src = fq + static_cast<size_t>(min(q, q_last)) * ND;
#pragma unroll
    for (int i = 0; i < ND; i++) { const uint32_t v = src[i]; nxt[i] = (q < n_quads) ? v : 0xFFFFFFFFu; }
  }
  int64_t acc = 0; int bad = 0;
  int cl[KK];   // this lane's picks per isoform: registers, not one LDS atomic per read (the G lanes of
#pragma unroll  // a chain hit the same K counters, which the LDS serialises)
  for (int k = 0; k < KK; k++) cl[k] = 0;
  for (int q0 = 0; q0 < nqw; q0 += G) {
    const int q = q0 + sub;
    uint32_t cur[ND];
#pragma unroll
    for (int i = 0; i < ND; i++) cur[i] = nxt[i];
    {
      const int qn = q + G;
      const uint32_t *src = fq + static_cast<size_t>(min(qn, q_last)) * ND;
#pragma unroll
      for (int i = 0; i < ND; i++) { const uint32_t v = src[i]; nxt[i] = (qn < n_quads) ? v : 0xFFFFFFFFu; }
    }
    const miso_u32x4 u = philox_gibbs<true>(rng, static_cast<uint32_t>(q), n0r0);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      // The reference's scan (miso_paired.c:11-22, 64-75) as compares against the running cumulative
      // weights: c_k = c_(k-1) + w_k with w_k = +0.0 for an incompatible isoform (bit-neutral: every weight
      // is >= +0), T = c_(KK-1), rnd = u T.  The scan stops at the first compatible k with `rnd < c_k` (two
      // compatible isoforms) / `!(rnd > c_k)` (more); the tests are monotone in k and an incompatible k
      // repeats its predecessor's c, so the pick is the NUMBER of isoforms whose test fails, held to
      // [first compatible, last compatible] (rnd = 0 or NaN before the first one; "ran off the end").
      uint32_t fr[KK]; double c[KK]; bool val[KK];
      double T = 0.0; int nv = 0, firstv = KK, lastv = -1;
#pragma unroll
      for (int k = 0; k < KK; k++) {
        const int idx = j * KK + k;
        fr[k] = (cur[idx >> 1] >> (16 * (idx & 1))) & 0xFFFFu;
        val[k] = fr[k] != FRAG_NONE;
        const double fpv = lds_fp[val[k] ? fr[k] : 0u];   // unconditional gather, masked below
        const double wk = ps[k] * fpv;
        T = T + (val[k] ? wk : 0.0);
        c[k] = T;
        nv += val[k] ? 1 : 0;
        firstv = (val[k] && firstv == KK) ? k : firstv;
        lastv = val[k] ? k : lastv;
      }
      const double rnd = miso_u01(u.v[j]) * T;
      const bool two = nv == 2;
      int sel = 0;
#pragma unroll
      for (int k = 0; k < KK; k++) {
        // bitwise, not short-circuit: `&&` on f64 compares comes back as branches
        const bool lt = rnd < c[k], gt = rnd > c[k];
        sel += ((two & !lt) | (!two & gt)) ? 1 : 0;
      }
      sel = max(sel, firstv);
      sel = min(sel, lastv);          // lastv = -1 (no compatible isoform: a padding read): not picked
      uint32_t fsel = 0;
#pragma unroll
      for (int k = 0; k < KK; k++) { fsel = (sel == k) ? fr[k] : fsel; cl[k] += (sel == k) ? 1 : 0; }
      const bool picked = sel >= 0;
      const int32_t v = stab[picked ? sel * il + static_cast<int>(fsel) : 0];
      const bool isbad = v == SFIX_BAD;
      bad |= (picked & isbad) ? 1 : 0;
      acc += (picked & !isbad) ? v : 0;
      if (WRITE) { if (write_ass && picked) drawass[4 * q + j] = static_cast<uint8_t>(sel); }
    }
  }
#pragma unroll
  for (int k = 0; k < KK; k++) if (cl[k]) atomicAdd(&cnt[k], cl[k]);
  acc_out = acc; bad_out = bad;
}


// ---- paired-end read loop over the DENSE records (device.hpp; host.cpp pack_event_masks) ----
// One lane = one quad of reads per trip, as pe_quads, but nothing in the loop asks whether an isoform is
// compatible: the record's byte for an incompatible (read, isoform) points at a probability of -0.0.
//   * per (read, isoform) two SDWA instructions on the record's byte: 8 x f (the LDS address of the
//     probability) and k il2 + f (the index of the score);
//   * cumulative weights c_k = c_(k-1) + psi_k fp[f_k] start from -0.0: the incompatible isoforms in
//     front of the first compatible one keep c_k = -0.0, every other c_k has exactly the bits of the
//     reference's running sum (adding -0.0 changes nothing; -0.0 + w = w, also for w = +0.0);
//   * the reference's scan (miso_paired.c:11-22, 64-75) stops at the first compatible k with `rnd < c_k`
//     (two compatible isoforms; the second one unconditionally) or `!(rnd > c_k)` (more).  As SIGNED
//     INTEGERS the bit patterns order like the doubles (all >= +0) with -0.0 below everything, so
//     "isoform k is passed over" is  bits(rnd) - adj >= bits(c_k),  adj = 0 / 1 for the two rules (rnd = +0
//     with adj = 1 gives -1: stops at the first compatible isoform, as `!(0 > c)` does);
//   * the tests are monotone in k (an incompatible isoform repeats its predecessor's c), so the pick is
//     the first k whose test holds: its score index by a chain of selects, the per-isoform counts from
//     the running totals "reads that passed over k" (no per-read counter update, no atomics in the loop);
//   * all of this is exact whenever rnd < T (then T > 0 and the scan cannot run past the last compatible
//     isoform).  Anything else -- T = 0, subnormal or non-finite weights -- takes pe_pick_exact, the
//     reference's scan as written (cold; MISO_PE_FORCE_EXACT=1 sends every read there, tests).
__device__ __attribute__((noinline)) int pe_pick_exact(const uint8_t *rec8, int K, int il2, const double *psi,
                                                       const double *fp_rep, bool two, uint32_t word) {
  // rec8[k] = the read's fragment-length index for isoform k, il2 - 2 (PE_ZERO) = incompatible: straight from the record
  // (no per-lane copy: a K-sized array indexed by a loop counter lives in scratch memory)
  const int zero = il2 - 2;
  double T = 0.0; int nv = 0;
  for (int k = 0; k < K; k++) {
    const int f = rec8[k];
    if (f != zero) { T = T + psi[k] * fp_rep[f]; nv++; }
  }
  const double rnd = miso_u01(word) * T;
  double cum = 0.0; int seen = 0, sel = -1;
  for (int k = 0; k < K; k++) {
    const int f = rec8[k];
    if (f == zero) continue;
    cum = cum + psi[k] * fp_rep[f];
    const bool stop = two ? (seen == 0 ? (rnd < cum) : true) : !(rnd > cum);
    seen++;
    if (sel < 0 && (stop || seen == nv)) sel = k;
  }
  return sel;
}

// LDS reads by byte address (the workgroup's dynamic LDS starts at address 0: sampler_grp checks): no
// pointer arithmetic on a symbol the compiler cannot fold, and never a flat_load.
typedef const __attribute__((address_space(3))) double *lds_cdp;
typedef const __attribute__((address_space(3))) int32_t *lds_cip;
__device__ __forceinline__ double lds_f64(uint32_t addr) { return *reinterpret_cast<lds_cdp>(static_cast<uintptr_t>(addr)); }
__device__ __forceinline__ int32_t lds_i32(uint32_t addr) { return *reinterpret_cast<lds_cip>(static_cast<uintptr_t>(addr)); }
// 8 x (byte B of w): one SDWA shift instead of shift + mask; (byte B of w) + add: one SDWA add
template <int B> __device__ __forceinline__ uint32_t byte_x8(uint32_t w, uint32_t three) {
  uint32_t r;
  if constexpr (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(w));
  else if constexpr (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(w));
  else if constexpr (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(w));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(w));
  return r;
}
template <int B> __device__ __forceinline__ uint32_t byte_plus(uint32_t w, uint32_t add) {
  uint32_t r;
  if constexpr (B == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(w), "s"(add));
  else if constexpr (B == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(w), "s"(add));
  else if constexpr (B == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(w), "s"(add));
  else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(w), "s"(add));
  return r;
}
// N (1..4) stopping tests of one read: isoform k + i is passed over when rb >= cb[i]; then the pick's table
// offset moves on to nx[i] and the isoform's running total grows by one.  The compares first, into N
// SGPR pairs, so that no select or add-with-carry waits on the compare in front of it.
template <int N>
__device__ __forceinline__ void pe_tests(int64_t rb, const int64_t *cb, const uint32_t *nx, uint32_t &fsel, int *over) {
  uint64_t m0, m1, m2, m3, junk;
  if constexpr (N == 4) {
    asm volatile(
        "v_cmp_ge_i64_e64 %5, %10, %11\n\tv_cmp_ge_i64_e64 %6, %10, %12\n\tv_cmp_ge_i64_e64 %7, %10, %13\n\tv_cmp_ge_i64_e64 %8, %10, %14\n\t"
        "v_cndmask_b32_e64 %0, %0, %15, %5\n\tv_addc_co_u32_e64 %1, %9, %1, 0, %5\n\t"
        "v_cndmask_b32_e64 %0, %0, %16, %6\n\tv_addc_co_u32_e64 %2, %9, %2, 0, %6\n\t"
        "v_cndmask_b32_e64 %0, %0, %17, %7\n\tv_addc_co_u32_e64 %3, %9, %3, 0, %7\n\t"
        "v_cndmask_b32_e64 %0, %0, %18, %8\n\tv_addc_co_u32_e64 %4, %9, %4, 0, %8\n\t"
        : "+v"(fsel), "+v"(over[0]), "+v"(over[1]), "+v"(over[2]), "+v"(over[3]), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(junk)
        : "v"(rb), "v"(cb[0]), "v"(cb[1]), "v"(cb[2]), "v"(cb[3]), "v"(nx[0]), "v"(nx[1]), "v"(nx[2]), "v"(nx[3]));
  } else if constexpr (N == 3) {
    asm volatile(
        "v_cmp_ge_i64_e64 %4, %8, %9\n\tv_cmp_ge_i64_e64 %5, %8, %10\n\tv_cmp_ge_i64_e64 %6, %8, %11\n\t"
        "v_cndmask_b32_e64 %0, %0, %12, %4\n\tv_addc_co_u32_e64 %1, %7, %1, 0, %4\n\t"
        "v_cndmask_b32_e64 %0, %0, %13, %5\n\tv_addc_co_u32_e64 %2, %7, %2, 0, %5\n\t"
        "v_cndmask_b32_e64 %0, %0, %14, %6\n\tv_addc_co_u32_e64 %3, %7, %3, 0, %6\n\t"
        : "+v"(fsel), "+v"(over[0]), "+v"(over[1]), "+v"(over[2]), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(junk)
        : "v"(rb), "v"(cb[0]), "v"(cb[1]), "v"(cb[2]), "v"(nx[0]), "v"(nx[1]), "v"(nx[2]));
  } else if constexpr (N == 2) {
    asm volatile(
        "v_cmp_ge_i64_e64 %3, %6, %7\n\tv_cmp_ge_i64_e64 %4, %6, %8\n\ts_nop 0\n\t"
        "v_cndmask_b32_e64 %0, %0, %9, %3\n\tv_addc_co_u32_e64 %1, %5, %1, 0, %3\n\t"
        "v_cndmask_b32_e64 %0, %0, %10, %4\n\tv_addc_co_u32_e64 %2, %5, %2, 0, %4\n\t"
        : "+v"(fsel), "+v"(over[0]), "+v"(over[1]), "=&s"(m0), "=&s"(m1), "=&s"(junk)
        : "v"(rb), "v"(cb[0]), "v"(cb[1]), "v"(nx[0]), "v"(nx[1]));
  } else {
    asm volatile(
        "v_cmp_ge_i64_e64 %2, %4, %5\n\ts_nop 1\n\t"
        "v_cndmask_b32_e64 %0, %0, %6, %2\n\tv_addc_co_u32_e64 %1, %3, %1, 0, %2\n\t"
        : "+v"(fsel), "+v"(over[0]), "=&s"(m0), "=&s"(junk)
        : "v"(rb), "v"(cb[0]), "v"(nx[0]));
  }
}
template <int KK>
__device__ __forceinline__ void pe_all_tests(int64_t rb, const int64_t (&cb)[KK], const uint32_t (&off)[KK], uint32_t &fsel, int (&over)[KK - 1]) {
  constexpr int NT = KK - 1;
#pragma unroll
  for (int k = 0; k + 4 <= NT; k += 4) pe_tests<4>(rb, &cb[k], &off[k + 1], fsel, &over[k]);
  constexpr int R = NT & 3, K0 = NT - R;
  if constexpr (R == 3) pe_tests<3>(rb, &cb[K0], &off[K0 + 1], fsel, &over[K0]);
  else if constexpr (R == 2) pe_tests<2>(rb, &cb[K0], &off[K0 + 1], fsel, &over[K0]);
  else if constexpr (R == 1) pe_tests<1>(rb, &cb[K0], &off[K0 + 1], fsel, &over[K0]);
}

// The loop.  fq: the event's records, n_quads of them followed by one quad of padding reads, which the lanes
// beyond the event's last quad process instead (it changes nothing: no range test in the loop).
// STAB_LDS: the score table is the chain's LDS copy at byte address stab_lds, else stab_glob in global memory.
template <int KK, int GT, bool WRITE, bool BADCHK, bool STAB_LDS>
__device__ __forceinline__ void pe_dense(const uint32_t *fq, const double *psi, const double *fp_rep,
                                         uint32_t stab_lds, const int32_t *stab_glob, int il2, int *dl, uint8_t *drawass,
                                         bool write_ass, int nqw, int n_quads, int n_draw, int sub,
                                         const GibbsRng &rng, uint32_t n0r0, bool force_exact,
                                         int64_t &acc_out, int &bad_out, int stride_rt = 0) {
  const int G = GT > 0 ? GT : stride_rt;   // lanes striding over the chain's quads (GT = 0: known at run time only -- chains on several workgroups)
  constexpr int ND = KK + 1;   // dwords per quad: 4 KK index bytes, one dword of flags
  double ps[KK];
#pragma unroll
  for (int k = 0; k < KK; k++) ps[k] = psi[k];
  int64_t acc = 0; int bad = 0;
  int over[KK - 1];   // reads of this lane that passed over isoform k
#pragma unroll
  for (int k = 0; k < KK - 1; k++) over[k] = 0;
  const uint32_t three = 3u;
  const unsigned char *stg = reinterpret_cast<const unsigned char *>(stab_glob);
  auto score = [&](uint32_t pk) __attribute__((always_inline)) {   // pk = k il2 + f: index into the score table
    if constexpr (STAB_LDS) return lds_i32(stab_lds + (pk << 2));
    else return *reinterpret_cast<const int32_t *>(stg + (pk << 2));
  };
  uint32_t krow[KK];   // k il2 (uniform)
#pragma unroll
  for (int k = 0; k < KK; k++) krow[k] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(k * il2));
  auto load = [&](uint32_t (&r)[ND], int q) __attribute__((always_inline)) {
    const uint32_t *src = fq + static_cast<uint32_t>(min(q, n_quads)) * static_cast<uint32_t>(ND);
#pragma unroll
    for (int i = 0; i < ND; i++) r[i] = src[i];
  };
  // The scores of a quad's four picks come from the event's table in global memory (a 64-byte line per pick).  Round 2 issued
  // the gathers when the picks were known -- at the END of the trip -- and added them one trip later, so that "nothing in the
  // loop waits for a gather it has just issued".  It did: the loop's header waits for vmcnt(0) (the next quad's records must
  // be there, and across the back edge the compiler counts conservatively), i.e. for the four gathers issued twenty
  // instructions earlier -- a full L2 / infinity-cache round trip per trip, the 0.43 of a wavefront's cycles that
  // SQ_WAIT_ANY shows (profiles/r05_wait_counters.txt).  Round 5 (MISO_PE_LATE_GATHER): a quad's gathers are issued at the
  // TOP of the NEXT trip, beside that trip's record loads, and added at the top of the trip after: whatever is outstanding
  // at the header is a whole trip old.  Four registers (the picks' table offsets) live across the back edge.
  // Up to MISO_PE_LATE_UPTO isoforms (same box, 7500 iterations: K = 5 948 -> 880 ms; K = 8 1079 -> 1105, K = 10 704 -> 724,
  // K = 16 1039 -> 1052 ms: from eight isoforms on the four registers cost more than the wait, profiles/r05_pe_late_gather.txt).
#ifndef MISO_PE_LATE_UPTO
#define MISO_PE_LATE_UPTO 6
#endif
#define MISO_PE_LATE_GATHER (KK <= MISO_PE_LATE_UPTO)
#ifndef MISO_PE_ASM_RECORDS
#define MISO_PE_ASM_RECORDS 1
#endif
#ifndef MISO_PE_ASM_RECORDS_UPTO
#define MISO_PE_ASM_RECORDS_UPTO 8
#endif
  int32_t pend[4] = {0, 0, 0, 0};
  auto settle = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int32_t v = pend[j];
      if (BADCHK) {
        const bool isbad = v == SFIX_BAD;
        bad |= isbad ? 1 : 0;
        acc += isbad ? 0 : v;
      } else {
#if MISO_PE_MAD
        uint64_t junk;
        asm("v_mad_i64_i32 %0, %1, %2, 1, %0" : "+v"(acc), "=s"(junk) : "v"(v));   // acc += v, one instruction
#else
        acc += v;
#endif
      }
    }
  };
  const uint32_t neutral = static_cast<uint32_t>(il2 - 1);        // PE_ONE of isoform 0: score 0
  uint32_t fsp[4] = {neutral, neutral, neutral, neutral};         // the previous quad's picks (table offsets), scores not yet fetched
  auto process = [&](const uint32_t (&cur)[ND], int q) __attribute__((always_inline)) {
    if constexpr (!STAB_LDS && MISO_PE_LATE_GATHER) {
      settle();                                                   // the quad before the previous one: fetched a trip ago
#pragma unroll
      for (int j = 0; j < 4; j++) pend[j] = score(fsp[j]);        // the previous quad's
    }
    const miso_u32x4 u = philox_gibbs<true>(rng, static_cast<uint32_t>(q), n0r0);
    const uint32_t flags = cur[KK];
    bool okj[4];
    uint32_t fs[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t pk[KK]; int64_t cb[KK];
      double T = -0.0;
#pragma unroll
      for (int k = 0; k < KK; k++) {
        const int h = j * KK + k;
        const uint32_t w = cur[h >> 2];
        uint32_t off;
        switch (h & 3) {   // h is a compile-time constant after unrolling
        case 0: off = byte_x8<0>(w, three); pk[k] = byte_plus<0>(w, krow[k]); break;
        case 1: off = byte_x8<1>(w, three); pk[k] = byte_plus<1>(w, krow[k]); break;
        case 2: off = byte_x8<2>(w, three); pk[k] = byte_plus<2>(w, krow[k]); break;
        default: off = byte_x8<3>(w, three); pk[k] = byte_plus<3>(w, krow[k]); break;
        }
        T = T + ps[k] * lds_f64(off);
        cb[k] = __double_as_longlong(T);
      }
      const double rnd = miso_u01(u.v[j]) * T;
      const bool ok = (rnd < T) & !force_exact;
      okj[j] = ok;
      int64_t rb = __double_as_longlong(rnd) - static_cast<int64_t>((flags >> j) & 1u);
      rb = ok ? rb : INT64_MIN;     // not exact here: pe_pick_exact below (and what this makes the tests do is taken back there)
      uint32_t fsel = pk[0];
#if MISO_PE_ASM_TESTS
      pe_all_tests<KK>(rb, cb, pk, fsel, over);
#else
#pragma unroll
      for (int k = 0; k < KK - 1; k++) {
        const bool pass = rb >= cb[k];
        over[k] += pass ? 1 : 0;
        fsel = pass ? pk[k + 1] : fsel;
      }
#endif
      if (WRITE) {
        if (write_ass && ok && 4 * q + j < n_draw) drawass[4 * q + j] = static_cast<uint8_t>(fsel / static_cast<uint32_t>(il2));
      }
      fs[j] = BADCHK ? (ok ? fsel : neutral) : fsel;
    }
    if constexpr (!STAB_LDS && MISO_PE_LATE_GATHER) {
#pragma unroll
      for (int j = 0; j < 4; j++) fsp[j] = fs[j];
    } else {
      if constexpr (!STAB_LDS) settle();   // the previous quad's
#pragma unroll
      for (int j = 0; j < 4; j++) pend[j] = score(fs[j]);
      if constexpr (STAB_LDS) settle();    // an LDS read is not worth the four registers across the trip (K = 3, 4: -4 %)
    }
    const bool okq = okj[0] & okj[1] & okj[2] & okj[3];
    if (__builtin_expect(__any(!okq), 0)) {
      // (the block's words again: kept alive for this cold path they were spilled to scratch in EVERY trip, 16 bytes per lane)
      const miso_u32x4 uc = philox_gibbs<true>(rng, static_cast<uint32_t>(q), n0r0);
#pragma unroll 1
      for (int j = 0; j < 4; j++) {
        if (okj[j]) continue;   // the reference's scan as written
        const uint8_t *rec8 = reinterpret_cast<const uint8_t *>(fq + static_cast<size_t>(min(q, n_quads)) * ND) + j * KK;
        const int sel = pe_pick_exact(rec8, KK, il2, psi, fp_rep, ((flags >> j) & 1u) == 0, uc.v[j]);
        // what the loop above did with rb = INT64_MIN: passed over the incompatible isoforms in front of the
        // first compatible one (their c is -0.0 = INT64_MIN) and took the score there
        int lead = 0;
        while (lead < KK - 1 && rec8[lead] == il2 - 2) lead++;
#pragma unroll
        for (int k = 0; k < KK - 1; k++) over[k] += ((k < sel) ? 1 : 0) - ((k < lead) ? 1 : 0);
        const int32_t vx = score(static_cast<uint32_t>(sel * il2 + rec8[sel]));
        if (BADCHK) {
          if (vx == SFIX_BAD) bad = 1; else acc += vx;
        } else {
          acc += static_cast<int64_t>(vx) - score(static_cast<uint32_t>(lead * il2 + rec8[lead]));
        }
        if (WRITE) { if (write_ass && 4 * q + j < n_draw) drawass[4 * q + j] = static_cast<uint8_t>(sel); }
      }
    }
  };
  if constexpr (KK <= MISO_PE_AB_UPTO) {
    // two quads per trip, each fetched while the other is worked on (no register copies)
    uint32_t A[ND], B[ND];
    int q = sub;
    load(A, q);
    for (int q0 = 0; q0 < nqw; q0 += 2 * G) {
      load(B, q + G);
      process(A, q);
      load(A, q + 2 * G);
      if (q0 + G < nqw) process(B, q + G);
      q += 2 * G;
    }
  } else if constexpr (!STAB_LDS && !MISO_PE_LATE_GATHER && MISO_PE_ASM_RECORDS && KK <= MISO_PE_ASM_RECORDS_UPTO) {
    // Seven and eight isoforms (round 5; same box: K = 7 1112 -> 991 ms, K = 8 1080 -> 1064; K = 10 702 -> 728, K = 16 1051 ->
    // 1106 ms: beyond eight the long unrolled body schedules worse around the fixed waits than the compiler's own, and the
    // header's wait stays): the NEXT quad's records are fetched by global loads the compiler does not see,
    // issued at the top of the trip and waited for -- written out -- at its end with vmcnt(4): the records have landed (they
    // are a trip old), the four score gathers issued just before may stay in flight.  The compiler then has no reason to put
    // vmcnt(0) at the loop's header (it did: the record registers were copied there), and the gathers are waited for where
    // they are added, a trip later.  (The late gather above does the same with four more registers across the back edge; here
    // they are not to be had.)  Every younger memory operation than the records -- the gathers, a spill, the cold path's loads
    // -- only makes vmcnt(4) wait longer: never too short.
    constexpr int NQ = ND / 4, NR = ND % 4;   // pieces: NQ x 16 bytes, then 8 and / or 4
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    u32x4_t r4[NQ > 0 ? NQ : 1]; u32x2_t r2 = {0u, 0u}; uint32_t r1 = 0u;
    auto issue = [&](int q) __attribute__((always_inline)) {
      const uint32_t *src = fq + static_cast<uint32_t>(min(q, n_quads)) * static_cast<uint32_t>(ND);
#pragma unroll
      for (int i = 0; i < NQ; i++) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r4[i]) : "v"(src + 4 * i) : "memory");
      if constexpr (NR >= 2) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r2) : "v"(src + 4 * NQ) : "memory");
      if constexpr (NR & 1) asm volatile("global_load_dword %0, %1, off" : "=v"(r1) : "v"(src + 4 * NQ + (NR & 2)) : "memory");
    };
    // (every piece is named IN the wait: nothing may read or move it before this point.  Round 6: with the 16-byte pieces named
    // by empty statements around the wait instead, a build without machine-level hoisting put a register copy of one between
    // the statement in front and the wait -- a read before the data has landed; `make check-isa` stopped that build)
    static_assert(NQ == 1 || NQ == 2, "records of five to eight isoforms");
    auto landed = [&](bool everything) __attribute__((always_inline)) {   // (a constant at both call sites)
      if constexpr (NQ == 2) {
        if (everything) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r4[0]), "+v"(r4[1]), "+v"(r2), "+v"(r1) : : "memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(r4[0]), "+v"(r4[1]), "+v"(r2), "+v"(r1) : : "memory");
      } else {
        if (everything) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r4[0]), "+v"(r2), "+v"(r1) : : "memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(r4[0]), "+v"(r2), "+v"(r1) : : "memory");
      }
    };
    issue(sub);
    landed(true);
    for (int q0 = 0; q0 < nqw; q0 += G) {
      uint32_t cur[ND];
#pragma unroll
      for (int i = 0; i < 4 * NQ; i++) cur[i] = r4[i >> 2][i & 3];
      if constexpr (NR >= 2) { cur[4 * NQ] = r2[0]; cur[4 * NQ + 1] = r2[1]; }
      if constexpr (NR & 1) cur[ND - 1] = r1;
      issue(q0 + sub + G);
      process(cur, q0 + sub);
      landed(false);
    }
  } else {
    // wider records: one loop body, the next quad fetched into a second set of registers and copied over
    // (two bodies cost more registers than the copies cost time: K = 5..16 were 10-15 % slower that way)
    uint32_t nxt[ND];
    load(nxt, sub);
    for (int q0 = 0; q0 < nqw; q0 += G) {
      uint32_t cur[ND];
#pragma unroll
      for (int i = 0; i < ND; i++) cur[i] = nxt[i];
      load(nxt, q0 + sub + G);
      process(cur, q0 + sub);
    }
  }
  if constexpr (!STAB_LDS && MISO_PE_LATE_GATHER) {   // drain: the last but one quad's scores, then the last quad's
    settle();
#pragma unroll
    for (int j = 0; j < 4; j++) pend[j] = score(fsp[j]);
    settle();
  } else if constexpr (!STAB_LDS) settle();   // the last quad's
#pragma unroll
  for (int k = 0; k < KK - 1; k++) if (over[k]) atomicAdd(&dl[k], over[k]);
  acc_out = acc; bad_out = bad;
}

}  // namespace

// WIDE (paired-end dense records, G = 64): ONE chain per workgroup -- genes with tens of thousands of read pairs, whose
// read loop on 16 or 32 lanes would outlast the rest of the batch (the reference costs O(reads) per gene and genes
// share nothing, miso_paired.c:393-552).  The four wavefronts keep the chain's state in their own slices and run the
// scalar step redundantly (same inputs, same routines, same bits); pe_dense deals the chain's quads over all 256
// lanes; the per-wavefront "passed over k" totals, score sums and bad flags meet through LDS (a.red_off: scratch
// behind the slices), two barriers per Gibbs step; wavefront 0 writes the outputs, every wavefront its share of the
// final picks.
template <int G, bool PE, int KC, bool WIDE = false>
#ifndef MISO_GRP_MINBLOCKS
#define MISO_GRP_MINBLOCKS 3   // measured: K=3 89.7k -> 114.4k events/s going from 2 to 3 (register budget 168)
#endif
// paired-end: workgroups per CU the register budget is set for.  Registers beat occupancy in pe_dense (K=5
// 30.3k -> 39.3k, K=8 20.1k -> 27.5k events/s going from 3 to 2; 4: 19.1k / 13.0k; K=3 53.1k -> 58.1k, K=4 47.1k ->
// 51.2k; one workgroup per CU: no gain up to 8 isoforms, 19.9k -> 12.5k at 10)
#ifndef MISO_GRP_PE_BLOCKS
#define MISO_GRP_PE_BLOCKS 2
#endif
__device__ __forceinline__ void grp_body(const KernelArgs &a, unsigned block_x) {   // workgroup block_x of the run
  constexpr int KLO = KC == 4 ? 3 : (KC == 8 ? 5 : (KC == 12 ? 9 : (KC == 16 ? 13 : 17)));   // the class holds K in [KLO, KC]
#ifndef MISO_GRP_MH_CH8
#define MISO_GRP_MH_CH8 8
#endif
  constexpr int MH_CH = PE ? (KC <= 8 ? (MISO_GRP_MH_CH8 < KC ? MISO_GRP_MH_CH8 : KC) : 4) : 1;   // chunk width of the Metropolis-Hastings step's serial chains (seq_sum_u)
  static_assert(!WIDE || (PE && G == 64), "workgroup-wide chains: paired-end, whole wavefronts");
  constexpr int GS = WIDE ? 0 : G;         // lanes striding over one chain's quads (WIDE: 256 x the chain's workgroups, known at run time)
  constexpr bool MH_ONE = PE && KC <= 4;   // a single pass without the loop around it (K=3 57.2k -> 60.6k; at five to eight isoforms the loop form is faster: 39.5k against 35.9k at K=5)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CPW = 64 / G;
  const int il2 = pe_dense_il2(a.il);
  const int fp_bytes = PE ? (((a.pe_dense ? a.pe_dense * il2 : a.il) * 8 + 15) & ~15) : 0;
  double *lds_fp = reinterpret_cast<double *>(smem);
  if (PE) {
    if (a.pe_dense) {   // one row per isoform: [probabilities, -0.0 (PE_ZERO), 1.0 (PE_ONE)]; row 0 serves the plain paths too
      for (int i = threadIdx.x; i < a.pe_dense * il2; i += blockDim.x) {
        const int f = i % il2;
        lds_fp[i] = f < a.il ? a.frag_prob[f] : (f == a.il ? -0.0 : 1.0);
      }
    } else {
      for (int i = threadIdx.x; i < a.il; i += blockDim.x) lds_fp[i] = a.frag_prob[i];
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef MISO_GRP_WAVETIME   // tools/wave_time.py mix: how long every chain's wavefront ran (100 MHz), through ChainStats::hw_id
  uint64_t wt_t0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt_t0) : : "memory");
#endif
  const int grp = lane / G, sub = lane - grp * G;
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  // WIDE: which chain this workgroup works on, alone or as one of several (coop.hpp)
  CoopGroup cg{0, 1, nullptr};
  long wide_slot = block_x;
  if (WIDE && a.coop_tab) {
    const int32_t *t = a.coop_tab + 4 * static_cast<size_t>(block_x);
    wide_slot = __builtin_amdgcn_readfirstlane(t[0]);
    cg.rank = __builtin_amdgcn_readfirstlane(t[1]); cg.n = __builtin_amdgcn_readfirstlane(t[2]);
    cg.mem = a.coop_mem + static_cast<size_t>(__builtin_amdgcn_readfirstlane(t[3])) * COOP_WORDS;
    if (a.coop_max_polls) cg.max_polls = a.coop_max_polls;
  }
  uint32_t coop_step = 0; bool coop_ok = true;
  const long wave_id = WIDE ? wide_slot : static_cast<long>(block_x) * 4 + wave;
  if (wave_id * CPW >= n_chains) return;  // no block-level barrier below (WIDE: the whole workgroup leaves)
  long slot = wave_id * CPW + grp;
  const bool live_all = slot < n_chains;
  const bool live = live_all && (!WIDE || (wave == 0 && cg.rank == 0));   // who stores the chain's outputs
  if (!live_all) slot = n_chains - 1;       // shadow a real chain, store nothing
  const int sub_r = WIDE ? cg.rank * 256 + static_cast<int>(threadIdx.x) : sub;   // this lane's place among the chain's lanes
  const int stride_r = WIDE ? 256 * cg.n : G;
#ifdef MISO_GRP_KS_FIXED   // experiment (with MISO_PE_ONLY_K): the slice layout of one isoform count at compile time
  const int ks = MISO_GRP_KS_FIXED, cs = a.cstride;
#else
  const int ks = a.kstride, cs = a.cstride;
#endif
  Slice S = carve(smem + fp_bytes + (static_cast<size_t>(wave) * CPW + grp) * grp_slice_bytes(ks, cs, a.tstride),
                  ks, cs);

  const int ev = a.slot_event[slot / a.C];
  const uint32_t chain = static_cast<uint32_t>(slot % a.C);
  const DevEvent E = a.events[ev];
  const int K = E.K;
  int Kw = K, nqw = (E.n_draw + 3) >> 2;    // wave-uniform loop bounds
  for (int off = 32; off >= 1; off >>= 1) {
    Kw = max(Kw, __shfl_xor(Kw, off));
    nqw = max(nqw, __shfl_xor(nqw, off));
  }
  // tell the compiler these bounds are wave-uniform: otherwise every loop over them becomes a
  // divergent loop with exec masking and an LDS wait per step (measured: 8x slower read loop)
  Kw = __builtin_amdgcn_readfirstlane(Kw);
  nqw = __builtin_amdgcn_readfirstlane(nqw);
  const uint32_t event_id = E.has_id ? E.explicit_id : a.first_event_id + static_cast<uint32_t>(ev);
  const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
  const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);
  for (int k0 = 0; k0 < Kw; k0 += G) {
    const int k = k0 + sub;
    if (k < K) {
      S.cst[k] = consts[k]; S.isc[k] = consts[K + k]; S.hm1[k] = consts[2 * K + k];
      S.alpha[k] = 0.0; S.psi[k] = 0.0; S.cnt[k] = 0; S.bas[k] = base[k]; S.dl[k] = 0;
    }
  }
  // single-end class path: usable when every chain of the wavefront has a class table
  const int n_dcls = PE ? 0 : E.n_dcls;
  const int n_units = PE ? 0 : E.n_units;
  const int n_pairs = PE ? 0 : E.n_pairs;
  int ncw = n_dcls, nuw = n_units, npw = n_pairs, tww = K - 1;
  bool cls_ok = !PE && cs > 0 && (n_dcls > 0 || E.n_draw == 0) && n_dcls <= cs;
  for (int off = 32; off >= 1; off >>= 1) {
    ncw = max(ncw, __shfl_xor(ncw, off));
    nuw = max(nuw, __shfl_xor(nuw, off));
    npw = max(npw, __shfl_xor(npw, off));
    tww = max(tww, __shfl_xor(tww, off));
  }
  cls_ok = __all(cls_ok);
  ncw = __builtin_amdgcn_readfirstlane(ncw);
  nuw = __builtin_amdgcn_readfirstlane(nuw);
  npw = __builtin_amdgcn_readfirstlane(npw);
  tww = __builtin_amdgcn_readfirstlane(tww);
  if (cls_ok) {   // the class tables stay in LDS for the whole run
    const uint32_t *gt = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_cls);
    const uint16_t *gp = reinterpret_cast<const uint16_t *>(a.in_pool + E.off_clsmask);
    for (int i = sub; i < CLS_WORDS * (n_dcls + 1); i += G) S.ctab[i] = gt[i];
    for (int k = sub; k < K; k += G) S.ctab[CLS_WORDS * (cs + 1) + k] = gt[CLS_WORDS * (n_dcls + 1) + k];   // A_k
    for (int i = sub; i < n_pairs; i += G) S.pairs[i] = gp[i];
  }
  wave_sync();
  Scalars c;
  c.lg_sum = consts[3 * K + 0]; c.lg_each = consts[3 * K + 1]; c.sigma = consts[3 * K + 2];
  c.sd = consts[3 * K + 3]; c.covar = consts[3 * K + 4];

  const uint32_t *masks = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_draw);
  const uint16_t *frags = reinterpret_cast<const uint16_t *>(a.in_pool + E.off_draw);
  const int32_t *sfix = reinterpret_cast<const int32_t *>(a.in_pool + E.off_sfix);
  // dense records (pe_dense): every event of the launch has them (runtime.hip)
  const bool dense = PE && a.pe_dense > 0;
  const uint32_t *dq = reinterpret_cast<const uint32_t *>(a.in_pool + (dense ? E.off_dense : E.off_draw));
  const int32_t *sfixd = reinterpret_cast<const int32_t *>(a.in_pool + (dense ? E.off_sfixd : E.off_sfix));
  if (PE && a.tstride > 0) {  // score table into the chain's slice: no per-read gather from L2
    if (dense) { for (int i = sub; i < K * il2; i += G) S.stab[i] = sfixd[i]; sfixd = S.stab; }
    else { for (int i = sub; i < K * a.il; i += G) S.stab[i] = sfix[i]; sfix = S.stab; }
    wave_sync();
  }
  const bool dense_nobad = dense && __all(E.dense_nobad != 0);
  // pe_dense reads the LDS by byte address: the dynamic LDS (the only LDS of this kernel) starts at 0
  if (dense && static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)) != 0u) __builtin_trap();
  const uint32_t stab_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(S.stab));
  const int32_t *sfixd_glob = reinterpret_cast<const int32_t *>(a.in_pool + (dense ? E.off_sfixd : E.off_sfix));   // provably global memory
  double *samples = reinterpret_cast<double *>(a.out_pool + E.off_samples);
  double *loglik = reinterpret_cast<double *>(a.out_pool + E.off_loglik);
  uint8_t *drawass = a.out_pool + E.off_drawass;
  int32_t *trace = (E.off_trace == NO_TRACE) ? nullptr
                                             : reinterpret_cast<int32_t *>(a.out_pool + E.off_trace);
  const int n_draw = E.n_draw, n_quads = (n_draw + 3) >> 2;
  int64_t rfix = 0; int rbad = 0;
  const GibbsRng rng = gibbs_rng_init(a.seed, event_id, chain);
  // paired-end fast path: one isoform count for the whole wavefront
  // (the K <= 32 class has the quad path up to 20 isoforms: BASELINE configs[3] is 3-20 per gene)
  // Chains of different K may share a wavefront (the last wavefront of every K inside a class): each
  // then runs its own instantiation of the quad loop under divergence -- the loop has no wavefront-wide
  // operation -- instead of dragging the whole wavefront to the generic path (those few wavefronts
  // used to take 3x as long as everything else in a mixed batch and set the kernel's duration).
  const bool pe_fast = PE && __all(K >= KLO && K <= (KC == 32 ? 20 : KC));
  if (WIDE && !(pe_fast && dense)) __builtin_trap();   // the host sends only genes with dense records here (runtime.hip)

#ifdef MISO_K2_PROFILE
  uint64_t gp_thr = 0, gp_loop = 0, gp_mh = 0;
#endif
  // ---- Gibbs step for the chain's current psi (in S.psi) ----
  // always_inline: outlined (as it was for the larger classes), the lambda sees the slices through generic
  // pointers and every LDS access becomes a flat_load / flat_store
  auto gibbs = [&](uint32_t iter, bool write_ass) __attribute__((always_inline)) {
    GPROF_T(t0);
    for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K) S.cnt[k] = 0; }
    wave_sync();
    bool slow = false;   // a non-final threshold of 2^32 cannot be held in 32 bits: direct path this time
    if (!PE && cls_ok && !__any(write_ass)) {
      // integer thresholds of every drawing-read class for the current psi: one (class, member)
      // pair per lane
      const int tw = ks - 1;
      // register-resident threshold set-up pays from ~9 isoforms on (measured: K=8 +6%, K=10 +6%,
      // K=16 +18%; K=3 -2%, K=5 -5% because the unroll is as wide as the class's largest K)
      // (the K <= 32 kernel has it for wavefronts of up to 20 isoforms: thresholds were more than
      // 40 % of the K = 18..20 iteration)
      constexpr bool REG_CAP = KC >= 12;
      constexpr int KR = KC == 32 ? 20 : KC;
      const bool reg_thr = REG_CAP && Kw <= KR;          // wave-uniform
      double psr[REG_CAP ? KR : 1];
      if constexpr (REG_CAP) {
        if (reg_thr) {
#pragma unroll
          for (int k = 0; k < KR; k++) psr[k] = (k < K) ? S.psi[k] : 0.0;
        }
      }
      for (int p0 = 0; p0 < npw; p0 += G) {
        const int p = p0 + sub;
        if (p < n_pairs) {
          const uint32_t pr = S.pairs[p];
          const int cc = static_cast<int>(pr >> 8), kp = static_cast<int>(pr & 0xFFu);
          const uint32_t m = S.ctab[CLS_WORDS * cc];
          double T = 0.0, cumw = 0.0;
          if (REG_CAP && reg_thr) {
            // psi from registers, branch-free: a serial chain of LDS reads (one per compatible
            // isoform, ~100 cycles each) was a third of the K=10 iteration.  Adding +0.0 for the
            // isoforms outside the class leaves the sum's bits unchanged (all terms are >= +0).
#pragma unroll
            for (int k = 0; k < (REG_CAP ? KR : 1); k++) {
              T = T + (((m >> k) & 1u) ? psr[k] : 0.0);   // ascending isoforms, as miso.c:11-22
              cumw = (k == kp) ? T : cumw;
            }
          } else {
            for (uint32_t mm = m; mm; mm &= mm - 1) {   // ascending isoforms, as miso.c:11-22
              const int kk = __ffs(mm) - 1;
              T = T + S.psi[kk];
              if (kk == kp) cumw = T;
            }
          }
          const double est = cumw * (4294967296.0 / T);
          const uint64_t t = (__popc(m) == 2) ? draw_threshold<false>(cumw, T, est) : draw_threshold<true>(cumw, T, est);
          S.thr[cc * tw + kp] = static_cast<uint32_t>(t);
          slow |= (t >> 32) != 0;
        }
      }
      wave_sync();
      // per class: running maximum over its members (first j with u < t_j == first j with
      // u < max(t_0..t_j)), spread over the isoform axis: thr[c][k] = threshold of the last member
      // <= k, 0 before the first member and from the last member on (those reads are in A_k)
      for (int c0 = 0; c0 < ncw; c0 += G) {
        const int cc = c0 + sub;
        if (cc < n_dcls) {
          const uint32_t m = S.ctab[CLS_WORDS * cc];
          const int kmax = 31 - __clz(static_cast<int>(m));
          uint32_t run = 0, val = 0;
          if (REG_CAP && reg_thr) {
            constexpr int KT = REG_CAP ? KR - 1 : 1;
            uint32_t tv[KT];   // all loads first: they are independent, the stores below alias them
#pragma unroll
            for (int k = 0; k < KT; k++) tv[k] = (k < K - 1 && ((m >> k) & 1u)) ? S.thr[cc * tw + k] : 0u;
#pragma unroll
            for (int k = 0; k < KT; k++) {
              if (k < K - 1) {
                if (k >= kmax) val = 0u;
                else if ((m >> k) & 1u) { run = tv[k] > run ? tv[k] : run; val = run; }
                S.thr[cc * tw + k] = val;
              }
            }
          } else {
            for (int k = 0; k < K - 1; k++) {
              if (k >= kmax) val = 0u;
              else if ((m >> k) & 1u) { const uint32_t t = S.thr[cc * tw + k]; run = t > run ? t : run; val = run; }
              S.thr[cc * tw + k] = val;
            }
          }
        }
      }
      for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K) S.dl[k] = 0; }
      wave_sync();
      GPROF_T(t1);
      GPROF_ADD(gp_thr, t0, t1);
      slow = __any(slow);
      if (!slow) {
        const uint32_t n0r0 = rng.p1hi ^ iter ^ rng.k0;
#define MISO_UNITS(TW)                                                                               \
  {                                                                                                  \
    int D[TW];                                                                                       \
    class_units<TW, G>(S.ctab, S.thr, tw, n_dcls, nuw, n_units, sub, rng, n0r0, D);                  \
    _Pragma("unroll") for (int j = 0; j < TW; j++) if (j < tw && D[j]) atomicAdd(&S.dl[j], D[j]);   \
  }
        if constexpr (KC == 4) { if (tww <= 2) MISO_UNITS(2) else MISO_UNITS(3) }
        else if constexpr (KC == 8) { if (tww <= 4) MISO_UNITS(4) else if (tww == 5) MISO_UNITS(5) else if (tww == 6) MISO_UNITS(6) else MISO_UNITS(7) }
        else if constexpr (KC == 12) { if (tww <= 9) MISO_UNITS(9) else MISO_UNITS(11) }
        else if constexpr (KC == 16) { MISO_UNITS(15) }
        else { if (tww <= 23) MISO_UNITS(23) else MISO_UNITS(31) }
#undef MISO_UNITS
        wave_sync();
        // D_k (+ the reads of classes that end at or before k) -> picks per isoform
        const uint32_t *A = S.ctab + CLS_WORDS * (cs + 1);
        for (int k0 = 0; k0 < Kw; k0 += G) {
          const int k = k0 + sub;
          if (k < K) {
            const int hi = (k < K - 1) ? S.dl[k] + static_cast<int>(A[k]) : n_draw;
            const int lo = (k > 0) ? S.dl[k - 1] + static_cast<int>(A[k - 1]) : 0;
            S.cnt[k] = hi - lo;
          }
        }
        wave_sync();
        GPROF_T(t2);
        GPROF_ADD(gp_loop, t1, t2);
        return;
      }
    }
    int64_t acc = 0; int bad = 0;
    if (PE && pe_fast && dense) {
      const uint32_t n0r0 = rng.p1hi ^ iter ^ rng.k0;
      for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K) S.dl[k] = 0; }
      wave_sync();
#define MISO_PED2(KK, LDS)                                                                            \
  {                                                                                                   \
    if (__any(write_ass)) pe_dense<KK, GS, true, true, LDS>(dq, S.psi, lds_fp, stab_lds, sfixd_glob, il2, S.dl, drawass, write_ass, nqw, n_quads, n_draw, sub_r, rng, n0r0, a.pe_force_exact != 0, acc, bad, stride_r); \
    else if (dense_nobad) pe_dense<KK, GS, false, false, LDS>(dq, S.psi, lds_fp, stab_lds, sfixd_glob, il2, S.dl, drawass, write_ass, nqw, n_quads, n_draw, sub_r, rng, n0r0, a.pe_force_exact != 0, acc, bad, stride_r); \
    else pe_dense<KK, GS, false, true, LDS>(dq, S.psi, lds_fp, stab_lds, sfixd_glob, il2, S.dl, drawass, write_ass, nqw, n_quads, n_draw, sub_r, rng, n0r0, a.pe_force_exact != 0, acc, bad, stride_r); \
  }
#ifdef MISO_PE_NOLOOP   // experiment: the kernel's registers without the read loop
#define MISO_PED(KK) { acc = n_draw; }
#else
#define MISO_PED(KK) { if (a.tstride > 0) MISO_PED2(KK, true) else MISO_PED2(KK, false) }
#endif
#ifdef MISO_PE_ONLY_K   // experiment: a kernel that carries ONE isoform count's loop (register budget of that count alone)
      MISO_PED(MISO_PE_ONLY_K)
#else
      if constexpr (KC == 4) { if (K == 3) MISO_PED(3) else MISO_PED(4) }
      else if constexpr (KC == 8) { if (K == 5) MISO_PED(5) else if (K == 6) MISO_PED(6) else if (K == 7) MISO_PED(7) else MISO_PED(8) }
      else if constexpr (KC == 12) { if (K == 9) MISO_PED(9) else if (K == 10) MISO_PED(10) else if (K == 11) MISO_PED(11) else MISO_PED(12) }
      else if constexpr (KC == 16) { if (K == 13) MISO_PED(13) else if (K == 14) MISO_PED(14) else if (K == 15) MISO_PED(15) else MISO_PED(16) }
      else { if (K == 17) MISO_PED(17) else if (K == 18) MISO_PED(18) else if (K == 19) MISO_PED(19) else MISO_PED(20) }
#endif
#undef MISO_PED2
#undef MISO_PED
      wave_sync();
#pragma unroll
      for (int off = G >> 1; off >= 1; off >>= 1) { acc += __shfl_xor(acc, off); bad |= __shfl_xor(bad, off); }
      if constexpr (WIDE) {   // the four wavefronts' totals -> every wavefront's own slice / registers
        struct Red { int64_t acc; int bad, pad; };
        Red *red = reinterpret_cast<Red *>(smem + a.red_off);
        if (lane == 0) red[wave] = Red{acc, bad, 0};
        __syncthreads();
        int tot = 0; int64_t ta = 0; int tb = 0;
        const size_t slice_bytes = grp_slice_bytes(ks, cs, a.tstride);
#pragma unroll
        for (int w = 0; w < 4; w++) {
          if (lane < K - 1) tot += carve(smem + fp_bytes + w * slice_bytes, ks, cs).dl[lane];
          ta += red[w].acc; tb |= red[w].bad;
        }
        __syncthreads();
        if (cg.n > 1) {   // ... and the chain's other workgroups' (coop.hpp): accumulator coop_step % 3, one barrier
          uint32_t *accp = coop_acc(cg, coop_step);
          if (wave == 0) {
            if (lane < K - 1 && tot) atomicAdd(&accp[4 + lane], static_cast<uint32_t>(tot));
            if (lane == 0) {
              atomicAdd(reinterpret_cast<unsigned long long *>(accp), static_cast<unsigned long long>(ta));
              if (tb) atomicOr(&accp[2], 1u);
            }
            if (cg.rank == 0 && lane < COOP_ACC) atomicExch(&coop_acc(cg, coop_step + 1)[lane], 0u);   // next step's, read last two steps ago
            __threadfence();   // this wavefront's atomics are at the L2 before the workgroup announces its arrival
          }
          int *flag = reinterpret_cast<int *>(smem + a.red_off + 64);
          coop_ok = coop_barrier(cg, coop_step, flag) && coop_ok;
          tot = lane < K - 1 ? static_cast<int>(coop_load(&accp[4 + lane])) : 0;
          ta = static_cast<int64_t>((static_cast<uint64_t>(coop_load(&accp[1])) << 32) | coop_load(&accp[0]));
          tb = static_cast<int>(coop_load(&accp[2]));
          coop_step++;
        }
        if (lane < K - 1) S.dl[lane] = tot;
        acc = ta; bad = tb;
        wave_sync();
      }
      // reads that passed over k - 1 but not k picked k
      for (int k0 = 0; k0 < Kw; k0 += G) {
        const int k = k0 + sub;
        if (k < K) S.cnt[k] = (k > 0 ? S.dl[k - 1] : n_draw) - (k < K - 1 ? S.dl[k] : 0);
      }
      wave_sync();
      rfix = E.base_sfix + acc;
      rbad = bad | E.base_bad;
      GPROF_T(t2);
      GPROF_ADD(gp_loop, t0, t2);
      return;
    }
    if (PE && pe_fast) {
      const uint32_t n0r0 = rng.p1hi ^ iter ^ rng.k0;
#define MISO_PEQ(KK)                                                                                  \
  {                                                                                                   \
    if constexpr ((KK) > MISO_PE_MASKED_FROM) pe_quads_masked<KK, G>(frags, S.psi, lds_fp, sfix, a.il, S.cnt, drawass, write_ass, nqw, n_quads, n_draw, sub, rng, n0r0, acc, bad); \
    else if (__any(write_ass)) pe_quads<KK, G, true>(frags, S.psi, lds_fp, sfix, a.il, S.cnt, drawass, write_ass, nqw, n_quads, n_draw, sub, rng, n0r0, acc, bad); \
    else pe_quads<KK, G, false>(frags, S.psi, lds_fp, sfix, a.il, S.cnt, drawass, write_ass, nqw, n_quads, n_draw, sub, rng, n0r0, acc, bad); \
  }
      // per-lane K: divergent only in wavefronts that mix isoform counts
#ifdef MISO_PE_ONLY_K
      MISO_PEQ(MISO_PE_ONLY_K)
#else
      if constexpr (KC == 4) { if (K == 3) MISO_PEQ(3) else MISO_PEQ(4) }
      else if constexpr (KC == 8) { if (K == 5) MISO_PEQ(5) else if (K == 6) MISO_PEQ(6) else if (K == 7) MISO_PEQ(7) else MISO_PEQ(8) }
      else if constexpr (KC == 12) { if (K == 9) MISO_PEQ(9) else if (K == 10) MISO_PEQ(10) else if (K == 11) MISO_PEQ(11) else MISO_PEQ(12) }
      else if constexpr (KC == 16) { if (K == 13) MISO_PEQ(13) else if (K == 14) MISO_PEQ(14) else if (K == 15) MISO_PEQ(15) else MISO_PEQ(16) }
      else { if (K == 17) MISO_PEQ(17) else if (K == 18) MISO_PEQ(18) else if (K == 19) MISO_PEQ(19) else MISO_PEQ(20) }
#endif
#undef MISO_PEQ
      wave_sync();
#pragma unroll
      for (int off = G >> 1; off >= 1; off >>= 1) { acc += __shfl_xor(acc, off); bad |= __shfl_xor(bad, off); }
      rfix = E.base_sfix + acc;
      rbad = bad | E.base_bad;
      return;
    }
    const bool small = KC <= 8;
    double ps[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ps[k] = (small && k < K) ? S.psi[k] : 0.0;
    for (int q0 = 0; q0 < nqw; q0 += G) {
      const int q = q0 + sub;
      const bool active = q < n_quads;
      const miso_u32x4 u = miso_draw_block(a.seed, event_id, chain, iter, MISO_SITE_GIBBS,
                                           static_cast<uint32_t>(q));
      uint32_t m4[4] = {0, 0, 0, 0};
      if (!PE && active) {
        const uint4 v = *reinterpret_cast<const uint4 *>(masks + 4 * static_cast<size_t>(q));
        m4[0] = v.x; m4[1] = v.y; m4[2] = v.z; m4[3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = 4 * q + j;
        const bool on = active && r < n_draw;
        int sel = -1; uint16_t fsel = 0;
        if constexpr (KC <= 8) {   // K <= 8: exact unroll for the wavefront's isoform count
          const uint16_t *row = frags + static_cast<size_t>(r) * K;
          if constexpr (KC == 4) {
            if (Kw <= 3) sel = pick_direct<3, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel);
            else sel = pick_direct<4, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel);
          } else {
            if (Kw <= 5) sel = pick_direct<5, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel);
            else if (Kw == 6) sel = pick_direct<6, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel);
            else if (Kw == 7) sel = pick_direct<7, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel);
            else sel = pick_direct<8, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel);
          }
        } else {
          double T = 0.0; int nv = 0;   // pass 1: total weight, ascending k (miso.c:11-22)
          for (int k = 0; k < Kw; k++) {
            if (PE) {
              const uint16_t f = (on && k < K) ? frags[static_cast<size_t>(r) * K + k] : FRAG_NONE;
              if (f != FRAG_NONE) { T = T + S.psi[k] * lds_fp[f]; nv++; }
            } else if (on && ((m4[j] >> k) & 1u)) { T = T + S.psi[k]; nv++; }
          }
          const double rnd = miso_u01(u.v[j]) * T;
          double cum = 0.0; int idx = 0;  // pass 2 (miso.c:69-80)
          for (int k = 0; k < Kw; k++) {
            bool valid; uint16_t f = 0;
            if (PE) {
              f = (on && k < K) ? frags[static_cast<size_t>(r) * K + k] : FRAG_NONE;
              valid = f != FRAG_NONE;
              if (valid) cum = cum + S.psi[k] * lds_fp[f];
            } else {
              valid = on && ((m4[j] >> k) & 1u);
              if (valid) cum = cum + S.psi[k];
            }
            if (valid) {
              const bool stop = (nv == 2) ? (idx == 0 ? (rnd < cum) : true) : !(rnd > cum);
              idx++;
              if (sel < 0 && (stop || idx == nv)) { sel = k; fsel = f; }
            }
          }
        }
        if (sel >= 0) {
          atomicAdd(&S.cnt[sel], 1);
          if (PE) {
            const int32_t v = sfix[static_cast<size_t>(sel) * a.il + fsel];
            if (v == SFIX_BAD) bad = 1; else acc += v;
          }
          if (write_ass) drawass[r] = static_cast<uint8_t>(sel);
        }
      }
    }
    wave_sync();
    if (PE) {
#pragma unroll
      for (int off = G >> 1; off >= 1; off >>= 1) { acc += __shfl_xor(acc, off); bad |= __shfl_xor(bad, off); }
      rfix = E.base_sfix + acc;
      rbad = bad | E.base_bad;
    }
  };
  auto count_of = [&](int k) { return S.bas[k] + S.cnt[k]; };

  // ---- alpha' = alpha + sd z ; psi' = logit_inv(alpha')  (miso.c:449-471): passes 1 (qnorm) and 2 (exp) ----
  auto propose = [&](double *alpha_in, double *alpha_out, double *psi_out,
                     uint32_t iter, uint32_t &accept_word) {
    {
      const miso_u32x4 b0 = miso_draw_block(a.seed, event_id, chain, iter, MISO_SITE_MH, 0u);
      accept_word = b0.v[0];
    }
    for (int k0 = 0; k0 < Kw; k0 += G) {
      const int k = k0 + sub;
      const int w = 2 + 2 * k;
      const miso_u32x4 b = miso_draw_block(a.seed, event_id, chain, iter, MISO_SITE_MH,
                                           static_cast<uint32_t>(w >> 2));
      if (k < K - 1) {
        const double z = miso_det_norm_from_unif(miso_u01(b.v[w & 3]), miso_u01(b.v[(w & 3) + 1]));
        const double an = alpha_in[k] + c.sd * z;
        alpha_out[k] = an;
        S.tc[k] = miso_det_exp(an);
      }
    }
    wave_sync();
    const double sumexp = seq_sum_u<MH_CH, MH_ONE>(S.tc, K - 1, Kw - 1) + 1.0;
    for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K - 1) psi_out[k] = S.tc[k] / sumexp; }
    wave_sync();
    const double sumpsi = seq_sum_u<MH_CH, MH_ONE>(psi_out, K - 1, Kw - 1);
    if (sub == 0) psi_out[K - 1] = 1 - sumpsi;
    wave_sync();
  };

  // ---- the psi-only part of both scores, cached per state: lp = log x, tb = lp + cst, lr =
  // log(x_k / x_K') with x_K' = 1 - x_0 - x_1 ... (miso.c:104-109), and the Jacobian 1/(prod x_i x_K').
  // ONE log pass over 2K-1 arguments. ----
  auto log_pass = [&](const double *x, double *lp, double *tb, double *lr, double &jac) {
    double ltheta = 1.0, prod = 1.0;
#pragma unroll 1
    for (int k0 = 0; k0 < (MH_ONE ? 1 : Kw - 1); k0 += MH_CH) {
      double t[MH_CH];
      load_chunk<MH_CH>(x, k0, K - 1, t);
#pragma unroll
      for (int i = 0; i < MH_CH; i++) { ltheta = ltheta - (k0 + i < K - 1 ? t[i] : 0.0); prod = prod * (k0 + i < K - 1 ? t[i] : 1.0); }
    }
    jac = 1.0 / prod / ltheta;
    for (int s0 = 0; s0 < 2 * Kw - 1; s0 += G) {
      const int s = s0 + sub;
      if (s < 2 * K - 1) {
        const bool first = s < K;
        const int k = first ? s : s - K;
        const double xv = x[k];
        const double r = miso_det_log(first ? xv : xv / ltheta);
        if (first) { lp[k] = r; tb[k] = r + S.cst[k]; } else { lr[k] = r; }
      }
    }
    wave_sync();
  };
  auto max_of = [&](const double *tb) { return seq_max_u<MH_CH, MH_ONE>(tb, K, Kw); };
  // ---- joint log score from cached logs and the current counts (miso.c:243-307; PE miso_paired.c:133-174) ----
  auto joint_sums = [&](const double *lp, const double *tb, double lse, double readProbPE) {
    double readProb = 0.0, assProb = 0.0, psiProb = 0.0;
#pragma unroll 1
    for (int k0 = 0; k0 < (MH_ONE ? 1 : Kw); k0 += MH_CH) {
      int ck[MH_CH]; double tbv[MH_CH], iscv[MH_CH], lpv[MH_CH], hv[MH_CH];
#pragma unroll
      for (int i = 0; i < MH_CH; i++) {
        const int kk = min(k0 + i, K - 1);
        ck[i] = count_of(kk); tbv[i] = tb[kk]; lpv[i] = lp[kk]; hv[i] = S.hm1[kk];
        iscv[i] = PE ? 0.0 : S.isc[kk];
      }
#pragma unroll
      for (int i = 0; i < MH_CH; i++) {
        const bool on = k0 + i < K && ck[i] != 0;
        if (!PE) readProb = on ? readProb + static_cast<double>(ck[i]) * iscv[i] : readProb;
        assProb = on ? assProb + static_cast<double>(ck[i]) * (tbv[i] - lse) : assProb;
      }
      // (the two sums are independent chains: interleaving them is the reference's order within each)
#pragma unroll
      for (int i = 0; i < MH_CH; i++) psiProb = k0 + i < K ? psiProb + hv[i] * lpv[i] : psiProb;
    }
    if (PE) readProb = readProbPE;
    psiProb = psiProb + c.lg_sum;
    psiProb = psiProb - c.lg_each;
    return readProb + assProb + psiProb;
  };

  // ---- initial state: miso.c:330-447 (START_AUTO / START_UNIFORM), then miso.c:834, 841 ----
  if (a.start == MISO_START_AUTO && K != 2)
    for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K - 1) S.alpha[k] = 1.0 / (K - 1); }
  wave_sync();
  uint32_t accept_word = 0;
  propose(S.alpha, S.alpha, S.psi, MISO_ITER_INIT, accept_word);
  double jac = 0.0, lse = 0.0;    // of the current psi
  {
    log_pass(S.psi, S.lp, S.tb, S.lr, jac);
    const double maxv = max_of(S.tb);
    for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K) S.tc[k] = miso_det_exp(S.tb[k] - maxv); }
    wave_sync();
    lse = miso_det_log(seq_sum_u<MH_CH, MH_ONE>(S.tc, K, Kw)) + maxv;
    wave_sync();
  }
  gibbs(MISO_ITER_INIT, live_all && chain == 0 && a.M == 0);

  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;

  RoundOpen ro(a);
  for (int m = 0; m < a.M; m++) {
    const bool opens = ro.at(a, m);   // a round's first iteration: no proposal terms (miso.c:866)
    prio_by_progress(a, m);
    if (WIDE && !coop_ok) return;   // the chain's workgroups gave up waiting for each other (coop.hpp): the host reports it
#pragma unroll 1
    for (int k0 = 0; k0 < (MH_ONE ? 1 : Kw); k0 += MH_CH) {
      int ck[MH_CH];
#pragma unroll
      for (int i = 0; i < MH_CH; i++) ck[i] = count_of(min(k0 + i, K - 1));
#pragma unroll
      for (int i = 0; i < MH_CH; i++) hash = k0 + i < K ? (hash ^ static_cast<uint32_t>(ck[i])) * 0x100000001B3ull : hash;
    }
    if (trace && live)
      for (int k0 = 0; k0 < Kw; k0 += G) {
        const int k = k0 + sub;
        if (k < K) trace[(static_cast<size_t>(m) * a.C + chain) * K + k] = count_of(k);
      }
    GPROF_T(m0);
    propose(S.alpha, S.alphaN, S.psiN, static_cast<uint32_t>(m), accept_word);     // passes 1, 2
    double jacN;
    log_pass(S.psiN, S.lpN, S.tbN, S.lrN, jacN);                                      // pass 3
    const double maxN = max_of(S.tbN);
    // Gaussian parts of the two proposal densities (miso.c:110-117): proposal -> current uses the
    // current psi's log ratios against alpha', current -> proposal the proposal's against alpha
    for (int k0 = 0; k0 < Kw; k0 += G) {
      const int k = k0 + sub;
      if (k < K - 1) {
        const double t1 = S.lr[k] - S.alphaN[k];
        S.tc[k] = (-0.5) * t1 * t1 / c.sigma;
        const double t2 = S.lrN[k] - S.alpha[k];
        S.u2[k] = (-0.5) * t2 * t2 / c.sigma;
      }
    }
    wave_sync();
    const double e1 = seq_sum_u<MH_CH, MH_ONE>(S.tc, K - 1, Kw - 1), e2 = seq_sum_u<MH_CH, MH_ONE>(S.u2, K - 1, Kw - 1);
    wave_sync();
    for (int s0 = 0; s0 < Kw + 2; s0 += G) {                                        // pass 4: exp
      const int s = s0 + sub;
      if (s < K + 2) {
        const double arg = s < K ? S.tbN[s] - maxN : (s == K ? e1 : e2);
        const double r = miso_det_exp(arg);
        if (s < K) S.tc[s] = r; else S.sx[s - K] = r;
      }
    }
    wave_sync();
    {                                                                               // pass 5: log
      const double sumtc = seq_sum_u<MH_CH, MH_ONE>(S.tc, K, Kw);
      const double x1 = S.sx[0], x2 = S.sx[1];
      wave_sync();
      for (int s0 = 0; s0 < 3; s0 += G) {
        const int s = s0 + sub;
        if (s < 3) S.sx[s] = miso_det_log(s == 0 ? sumtc : (s == 1 ? c.covar * jac * x1 : c.covar * jacN * x2));
      }
      wave_sync();
    }
    const double lseN = S.sx[0] + maxN, ptoCS = S.sx[1], ctoPS = S.sx[2];
    const double rp = PE ? (rbad ? miso_u2d(0x7FF8000000000000ull)
                                 : static_cast<double>(rfix) * (1.0 / MISO_SFIX_SCALE))
                         : 0.0;
    const double pp = joint_sums(S.lpN, S.tbN, lseN, rp);
    const double pc = joint_sums(S.lp, S.tb, lse, rp);
    const double acceptP = !opens ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);  // pass 6
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    double cJS = pc;
    wave_sync();
    if (acc) {   // the proposal and its cache become the current state: swap the slice's pointers
      double *t;
      t = S.psi; S.psi = S.psiN; S.psiN = t;
      t = S.alpha; S.alpha = S.alphaN; S.alphaN = t;
      t = S.lp; S.lp = S.lpN; S.lpN = t;
      t = S.tb; S.tb = S.tbN; S.tbN = t;
      t = S.lr; S.lr = S.lrN; S.lrN = t;
      jac = jacN; lse = lseN;
      cJS = pp; accepted++;
    }
    GPROF_T(m1);
    GPROF_ADD(gp_mh, m0, m1);
#ifndef MISO_GRP_STORE_AFTER
#define MISO_GRP_STORE_AFTER 0
#endif
    // miso.c:882-893.  (MISO_GRP_STORE_AFTER: the stores issued behind the Gibbs step, whose first record load is only "there"
    // once the stores in front of it have been acknowledged -- vmcnt counts both, in order.  The Gibbs step does not touch psi:
    // the same values leave either way.  Tried; not the wait.)
    bool rec = false; size_t rec_col = 0;
    if (m >= a.B) {
      if (lagCounter == a.lag - 1) {
        rec = live; rec_col = static_cast<size_t>(noS) + chain;
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    auto store_sample = [&]() __attribute__((always_inline)) {
      for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K) samples[rec_col * K + k] = S.psi[k]; }
      if (sub == 0) loglik[rec_col] = cJS;
    };
    if (!MISO_GRP_STORE_AFTER && rec) store_sample();
    gibbs(static_cast<uint32_t>(m), live_all && chain == 0 && m == a.M - 1);
    if (MISO_GRP_STORE_AFTER && rec) store_sample();
  }
  for (int k = 0; k < K; k++) hash = (hash ^ static_cast<uint32_t>(count_of(k))) * 0x100000001B3ull;
#ifdef MISO_K2_PROFILE
  if (live && sub == 0 && chain == 0 && a.M > 8) {
    loglik[0] = static_cast<double>(gp_mh); loglik[1] = static_cast<double>(gp_thr); loglik[2] = static_cast<double>(gp_loop);
  }
#endif
  if (live) {
    if (trace)
      for (int k0 = 0; k0 < Kw; k0 += G) {
        const int k = k0 + sub;
        if (k < K) trace[(static_cast<size_t>(a.M) * a.C + chain) * K + k] = count_of(k);
      }
    if (sub == 0) {
      ChainStats *st = reinterpret_cast<ChainStats *>(a.out_pool + E.off_stats) + chain;
      st->counts_hash = hash; st->accepted = accepted; st->hw_id = 0;
#ifdef MISO_GRP_WAVETIME
      uint64_t wt_t1;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt_t1) : : "memory");
      st->hw_id = static_cast<uint32_t>(wt_t1 - wt_t0);
      st->counts_hash = wt_t0;   // when it started
#endif
    }
  }
}

template <int G, bool PE, int KC, bool WIDE = false>
__global__ __launch_bounds__(256, PE ? MISO_GRP_PE_BLOCKS : (KC <= 8 ? MISO_GRP_MINBLOCKS : 2)) void sampler_grp(const KernelArgs a) {
  grp_body<G, PE, KC, WIDE>(a, blockIdx.x);
}

// Paired-end, one isoform-count class with its size buckets in ONE launch (runtime.hip): runs of the launch's genes --
// one chain per workgroup (with the chains on several workgroups in front), per wavefront, on 32 and on 16 lanes --
// as a.n_segs segments: workgroups [seg_block[s], seg_block[s + 1]), genes (slots) [seg_slot[s], seg_slot[s + 1]) of
// the launch's list, seg_lanes[s] lanes per chain (K2_WIDE = a workgroup).  One kernel in the hardware queue instead of
// four, the workgroups start longest chains first across the buckets (docs/history.md 4.3 (iv)).
template <int KC>
__device__ __forceinline__ void grp_lanes(const KernelArgs &b, unsigned blk, int lanes) {
  switch (lanes) {
  case K2_WIDE: grp_body<64, true, KC, true>(b, blk); break;
  case 64: grp_body<64, true, KC, false>(b, blk); break;
  case 32: grp_body<32, true, KC, false>(b, blk); break;
  case 8: grp_body<8, true, KC, false>(b, blk); break;   // the class's small genes (round 6)
  default: grp_body<16, true, KC, false>(b, blk); break;
  }
}
template <int KC>
__global__ __launch_bounds__(256, MISO_GRP_PE_BLOCKS) void sampler_grp_multi(const KernelArgs a) {
  int s = 0;
  while (s + 1 < a.n_segs && static_cast<int>(blockIdx.x) >= a.seg_block[s + 1]) s++;
  s = __builtin_amdgcn_readfirstlane(s);
  KernelArgs b = a;
  b.slot_event = a.slot_event + a.seg_slot[s];
  b.n_slots = a.seg_slot[s + 1] - a.seg_slot[s];
  b.tstride = a.seg_ts[s];
  const unsigned blk = blockIdx.x - static_cast<unsigned>(a.seg_block[s]);
  grp_lanes<KC>(b, blk, a.seg_lanes[s]);
}

// Paired-end, EVERY isoform-count class of a batch of like-sized genes (sixteen lanes per chain everywhere) in one launch
// (runtime.hip; VERDICT r5 item 5: a whole-gene batch as one ordered grid).  The segments (a.grp_segs, in global memory: a run-time
// index into the by-value arguments would cost a scratch copy of them) are the classes' runs ordered by what a workgroup of
// theirs costs: the hardware starts workgroups in index order, so the 17 - 20 isoform genes all start in the first round of
// resident wavefronts and the three-isoform genes fill the slots they free -- instead of five kernels in five hardware queues
// whose workgroups the dispatcher interleaves as it likes (profiles/r06_mix_timeline.txt).
// Sixteen lanes only.  The other kinds of run (32 lanes, a wavefront, a workgroup or several per chain: the size buckets of real
// pair counts) were built the same way, a kernel per kind with every class's runs cut into pieces ordered by cost: 825 -> 1010 ms
// on hg19-like pair counts, with whole runs or pieces alike -- the launch per class (sampler_grp_multi) stays there; and all
// twenty bodies in ONE function took 28 minutes to compile and spilled 802 registers (round 5).
#ifdef MISO_GRP_ALL_CLASSES
__global__ __launch_bounds__(256, MISO_GRP_PE_BLOCKS) void sampler_grp_all(const KernelArgs a) {
  int s = 0;
  while (s + 1 < a.n_grp_segs && static_cast<int>(blockIdx.x) >= a.grp_segs[s + 1].block0) s++;
  s = __builtin_amdgcn_readfirstlane(s);
  // (the table is read by vector loads -- nothing tells the compiler that it is not written meanwhile --: every field is made a
  // scalar by hand, or the slice layout and the loop bounds of the body would count as different from lane to lane)
  const GrpSeg *gp = a.grp_segs + s;
  const int block0 = __builtin_amdgcn_readfirstlane(gp->block0), kc = __builtin_amdgcn_readfirstlane(gp->kc);
  KernelArgs b = a;
  b.slot_event = a.slot_event + __builtin_amdgcn_readfirstlane(gp->slot0);
  b.n_slots = __builtin_amdgcn_readfirstlane(gp->n_slots);
  b.kstride = __builtin_amdgcn_readfirstlane(gp->kstride); b.tstride = __builtin_amdgcn_readfirstlane(gp->tstride);
  const unsigned blk = blockIdx.x - static_cast<unsigned>(block0);
  switch (kc) {
  case 4: grp_body<16, true, 4, false>(b, blk); break;
  case 8: grp_body<16, true, 8, false>(b, blk); break;
  case 12: grp_body<16, true, 12, false>(b, blk); break;
  case 16: grp_body<16, true, 16, false>(b, blk); break;
  default: grp_body<16, true, 32, false>(b, blk); break;
  }
}
#endif

}  // namespace miso

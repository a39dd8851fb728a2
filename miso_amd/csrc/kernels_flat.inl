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

#ifndef MISO_FLAT_UQ
#define MISO_FLAT_UQ 2   // Philox blocks (work units) per lane and trip of the read loop (4 measured slower: K=10 39.1k vs 41.1k)
#endif

#ifdef MISO_K2_PROFILE
#define FPROF_T(var) const uint64_t var = __builtin_readcyclecounter()
#define FPROF_ADD(acc, t0, t1) acc += (t1) - (t0)
#else
#define FPROF_T(var)
#define FPROF_ADD(acc, t0, t1)
#endif

namespace miso {

// the kernel's dynamic LDS (declared here so that flat_units, which is not inlined, addresses it as LDS too)
extern __shared__ __attribute__((aligned(16))) unsigned char smem_flat[];

namespace {

// LDS traffic between lanes of ONE wavefront: program order is enough once the compiler may not move
// the accesses (no s_barrier: a chain never spans wavefronts)
__device__ __forceinline__ void fsync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// indices into a chain's double scalars (FlatLayout::sx) ...
enum { SX_SUMEXP = 0, SX_LTHETA, SX_MAXV, SX_E1, SX_E2, SX_X1, SX_X2, SX_LA0, SX_LA1, SX_LA2, SX_LR0, SX_LR1,
       SX_LR2, SX_SD, SX_SIGMA, SX_COVAR };
// ... and int scalars (FlatLayout::misc)
enum { MI_K = 0, MI_NDRAW, MI_NCLS, MI_NUNITS, MI_EVID, MI_CHAIN, MI_ACC, MI_ACCW, MI_C3K1, MI_P1LO, MI_P1HIK0,
       MI_EV, MI_SAMP_LO, MI_SAMP_HI, MI_TRACE_LO, MI_TRACE_HI, MI_USTART, MI_NEXT, MI_LANE0, MI_LANES, MI_DESC,
       MI_SLOW };   // MI_SLOW: one of the chain's current thresholds does not fit 32 bits (the direct path until psi changes)
static_assert(MI_SLOW < FLAT_MISC, "FlatLayout::misc");

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
// The same count when T is a normal finite number and 2 <= est <= 2^32 - 3 (the caller checks; otherwise it
// takes flat_threshold): est is then within 2^-20 of the boundary (one rounding in c / T 2^32, one in the
// test's product), so none of flat_threshold's clamps, range tests or outer tests can fire, and the count
// is floor(est) - 1 + (tests that hold among floor - 1, floor, floor + 1) -- two tests decide it, because
// the test is monotone in u: if it holds at floor the count is floor + 1 or floor + 2, else floor - 1 or floor.
__device__ __forceinline__ double flat_threshold_fast(bool le, double c, double T, double est) {
  const double t0 = __builtin_floor(est);
  auto pred = [&](double u) { const double rnd = u * (1.0 / 4294967296.0) * T; return le ? !(rnd > c) : (rnd < c); };
  const bool p0 = pred(t0);
  const double u1 = p0 ? t0 + 1.0 : t0 - 1.0;
  const bool p1 = pred(u1);
  return p0 ? (p1 ? t0 + 2.0 : t0 + 1.0) : (p1 ? t0 : t0 - 1.0);
}

// D += (w0 < T) + (w1 < T) + (w2 < T) + (w3 < T).  Written out: hipcc funnels every compare through VCC
// (v_cmp, two wait states, v_addc), three issue slots per compare; with four SGPR pairs in flight the
// wait states are covered by the neighbouring compares: two slots per compare.
__device__ __forceinline__ void count_below(int &D, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t T) {
  uint64_t m0, m1, m2, m3, junk;
  asm volatile(
      "v_cmp_lt_u32_e64 %1, %6, %10\n\t"
      "v_cmp_lt_u32_e64 %2, %7, %10\n\t"
      "v_cmp_lt_u32_e64 %3, %8, %10\n\t"
      "v_cmp_lt_u32_e64 %4, %9, %10\n\t"
      "v_addc_co_u32_e64 %0, %5, %0, 0, %1\n\t"
      "v_addc_co_u32_e64 %0, %5, %0, 0, %2\n\t"
      "v_addc_co_u32_e64 %0, %5, %0, 0, %3\n\t"
      "v_addc_co_u32_e64 %0, %5, %0, 0, %4\n\t"
      : "+v"(D), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(junk)
      : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(T));
}

// The read loop of one wavefront-iteration for at most TW + 1 isoforms per chain (TW = the launch's
// flat_trow or narrower; thr rows are zero beyond a chain's K - 1).
// Lane state: chain s, class c, unit i (within the chain), the class's unit range and thresholds -- and
// the SAME for the class that follows (N...), loaded when the current one was entered: by the time the
// lane gets there the LDS reads are long done, so a class change costs register moves, not a chain of
// dependent LDS round trips in front of the whole wavefront (some lane changes class in almost every
// trip; profiles/r02_flat_phase_cycles_v1.txt).  Not inlined: its registers are allocated on their own,
// away from the scalar pressure of the kernel's other phases.
struct FlatUnitsArgs {
  int woff;   // the wavefront's slices start here in smem_flat
  int slice, off_ctab, off_thr, off_misc, off_dl, trow, trips;
  uint32_t iter, k0, k1;
};
template <int TW>
__device__ __attribute__((noinline)) void flat_units(const FlatUnitsArgs A, int s0, int c0, int i0, int n_mine) {
  if (A.trips == 0) return;
  unsigned char *wbase = smem_flat + A.woff;
  const int slice = __builtin_amdgcn_readfirstlane(A.slice), off_ctab = __builtin_amdgcn_readfirstlane(A.off_ctab),
            off_thr = __builtin_amdgcn_readfirstlane(A.off_thr), off_misc = __builtin_amdgcn_readfirstlane(A.off_misc),
            off_dl = __builtin_amdgcn_readfirstlane(A.off_dl), trow4 = 4 * __builtin_amdgcn_readfirstlane(A.trow);
  const uint32_t iter = __builtin_amdgcn_readfirstlane(A.iter);
  // current class: unit range, thresholds; current chain: slice, classes left, next chain with units
  int i = i0, uend = i0, ust = 0, qd = 0, left = 0, next = -1;   // uend == i: the first trip enters a class
  int sl = 0;   // byte offset of the chain's slice
  uint32_t hm = 0xFFu;
  uint32_t T[TW], NT[TW];
  int D[TW];
#pragma unroll
  for (int j = 0; j < TW; j++) { T[j] = 0u; NT[j] = 0u; D[j] = 0; }
  GibbsRng rng;
  rng.k0 = A.k0; rng.k1 = A.k1; rng.p1lo = 0; rng.p1hi = 0; rng.c3k1 = 0;
  uint32_t n0r0 = 0;
  // the class after the current one (loaded when the current one was entered)
  int N_new = 1;        // ... is the first of another chain (or the lane's very first)
  int N_sl = 0, N_rowp = 0, N_thp = 0, N_i = i0, N_ust = 0, N_uend = 0, N_qd = 0, N_left = 0, N_next = -1;
  uint32_t N_hm = 0xFFu, N_c3k1 = 0, N_p1lo = 0, N_p1hik0 = 0;
  int rowp = 0, thp = 0;   // byte offsets of the current class's table row / threshold row
  bool have = false;       // D holds counts of the chain at sl
  auto prefetch = [&](int psl, int prow, int pth, int newchain) __attribute__((always_inline)) {
    const uint32_t *row = reinterpret_cast<const uint32_t *>(wbase + prow);
    const uint32_t *th = reinterpret_cast<const uint32_t *>(wbase + pth);
    N_ust = static_cast<int>(row[1]); N_qd = static_cast<int>(row[2]); N_hm = row[3];
    N_uend = static_cast<int>(row[CLS_WORDS + 1]);
#pragma unroll
    for (int j = 0; j < TW; j++) NT[j] = th[j];
    if (newchain) {
      const int *mi = reinterpret_cast<const int *>(wbase + psl + off_misc);
      N_left = mi[MI_NCLS]; N_next = mi[MI_NEXT];
      N_c3k1 = static_cast<uint32_t>(mi[MI_C3K1]); N_p1lo = static_cast<uint32_t>(mi[MI_P1LO]);
      N_p1hik0 = static_cast<uint32_t>(mi[MI_P1HIK0]);
    }
    N_sl = psl; N_rowp = prow; N_thp = pth; N_new = newchain;
  };
  if (n_mine > 0) {
    const int psl = s0 * slice;
    prefetch(psl, psl + off_ctab + 4 * CLS_WORDS * c0, psl + off_thr + c0 * trow4, 1);
    N_left -= c0;   // classes of the first chain from c0 on
  }
  // Up to UQ units per trip -- as many as the class and the lane's range still hold (blocks beyond that are
  // masked out): the class bookkeeping, the edge masks and the loop control are paid once per trip, and the
  // Philox blocks of a trip interleave.  Measured per block and wavefront at UQ = 2: generator 36 +
  // compares 8 (K - 1) + ~58 of bookkeeping (profiles/r02_flat_phase_valu.txt); UQ = 4 halves the latter per
  // block but masks out more blocks at class ends and range ends: measured 3-5 % slower at every K.
  constexpr int UQ = TW <= 15 ? MISO_FLAT_UQ : 2;   // (the widest rows keep two: registers)
  int rem = n_mine;   // units this lane still owes
  for (;;) {
    if (!__any(rem > 0)) break;
    const bool active = rem > 0;
    if (active && i == uend) {   // enter the prefetched class
      if (N_new) {
        if (have) {
          int *dl = reinterpret_cast<int *>(wbase + sl + off_dl);
#pragma unroll
          for (int j = 0; j < TW; j++) { atomicAdd(&dl[j], D[j]); D[j] = 0; }
        }
        sl = N_sl; left = N_left; next = N_next; i = N_i; N_i = 0;
        rng.c3k1 = N_c3k1; rng.p1lo = N_p1lo; n0r0 = N_p1hik0 ^ iter;
        have = true;
      }
      rowp = N_rowp; thp = N_thp; ust = N_ust; uend = N_uend; qd = N_qd; hm = N_hm;
#pragma unroll
      for (int j = 0; j < TW; j++) T[j] = NT[j];
      left--;
      if (left > 0) prefetch(sl, rowp + 4 * CLS_WORDS, thp + trow4, 0);
      else if (next >= 0) { const int psl = next * slice; prefetch(psl, psl + off_ctab, psl + off_thr, 1); }
    }
    // blocks of this trip: units i .. i + nb - 1 of the current class
    const int nb = active ? min(min(UQ, rem), uend - i) : 0;
    uint32_t w[UQ][4];
#pragma unroll
    for (int b = 0; b < UQ; b++) {
      const miso_u32x4 u = philox_gibbs<true>(rng, static_cast<uint32_t>(i + b - qd), n0r0);
      // words of a unit that belong to the class: all four except in the class's first / last unit; a word
      // outside becomes 0xFFFFFFFF (never below a 32-bit threshold)
      uint32_t wm = (b < nb) ? 0xFu : 0u;
      if (b == 0) wm &= (i == ust) ? hm : 0xFu;
      wm &= (i + b == uend - 1) ? (hm >> 4) : 0xFu;
      const int nm = static_cast<int>(~wm);
#pragma unroll
      for (int x = 0; x < 4; x++) w[b][x] = u.v[x] | static_cast<uint32_t>(__builtin_amdgcn_sbfe(nm, x, 1));
    }
#pragma unroll
    for (int j = 0; j < TW; j++) {
#pragma unroll
      for (int b = 0; b < UQ; b++) count_below(D[j], w[b][0], w[b][1], w[b][2], w[b][3], T[j]);
    }
    i += nb; rem -= nb;
  }
  if (have) {
    int *dl = reinterpret_cast<int *>(wbase + sl + off_dl);
#pragma unroll
    for (int j = 0; j < TW; j++) atomicAdd(&dl[j], D[j]);
  }
}


// The read loop over UNIT DESCRIPTORS (device.hpp DevEvent::off_units, host.cpp): every chain of the wavefront
// has a fixed group of lanes, in proportion to its units (sampler_flat's set-up); lane r of g takes the chain's
// units r, r + g, r + 2g ...: the descriptor (class, Philox block, which of its four words belong to the class)
// comes from global memory, coalesced within the group and fetched one trip ahead, the class's thresholds from
// the chain's LDS row.  No walk over classes and chains, no state to carry from unit to unit: against flat_units
// (contiguous unit ranges per lane, perfectly balanced) the lane groups are rounded to whole lanes (a few per
// cent more trips) and a trip costs ~30 % fewer instructions.
#ifndef MISO_FLAT_DESC_UQ
#define MISO_FLAT_DESC_UQ 2   // units per lane and trip
#endif
template <int TW>
__device__ __attribute__((noinline)) void flat_units_desc(const FlatUnitsArgs A, const uint32_t *pool_words, int ms, int r, int g) {
  if (A.trips == 0 || g <= 0) return;
  unsigned char *wbase = smem_flat + A.woff;
  const int slice = __builtin_amdgcn_readfirstlane(A.slice), off_thr = __builtin_amdgcn_readfirstlane(A.off_thr),
            off_misc = __builtin_amdgcn_readfirstlane(A.off_misc), off_dl = __builtin_amdgcn_readfirstlane(A.off_dl),
            trow4 = 4 * __builtin_amdgcn_readfirstlane(A.trow);
  const uint32_t iter = __builtin_amdgcn_readfirstlane(A.iter);
  const int sl = ms * slice;
  const int *mi = reinterpret_cast<const int *>(wbase + sl + off_misc);
  const int nu = mi[MI_NUNITS];
  const uint32_t *desc = pool_words + static_cast<uint32_t>(mi[MI_DESC]);
  GibbsRng rng;
  rng.k0 = A.k0; rng.k1 = A.k1; rng.p1hi = 0;
  rng.c3k1 = static_cast<uint32_t>(mi[MI_C3K1]); rng.p1lo = static_cast<uint32_t>(mi[MI_P1LO]);
  const uint32_t n0r0 = static_cast<uint32_t>(mi[MI_P1HIK0]) ^ iter;
  const unsigned char *thr = wbase + sl + off_thr;
  constexpr int UQ = MISO_FLAT_DESC_UQ;
  int D[TW];
#pragma unroll
  for (int j = 0; j < TW; j++) D[j] = 0;
#ifndef MISO_FLAT_ASM_PREFETCH
#define MISO_FLAT_ASM_PREFETCH 1
#endif
  uint32_t dn[UQ];   // the next trip's descriptors (0 = no unit: no word counts)
#if MISO_FLAT_ASM_PREFETCH
  // The descriptors are fetched one trip ahead -- and until round 5 were waited for at once: `desc` is a generic pointer in
  // this (non-inlined) function, a flat_load may return out of order with anything, so the compiler put s_waitcnt vmcnt(0)
  // lgkmcnt(0) between the prefetch it had just issued and the first use of the CURRENT descriptors -- a global-memory round
  // trip per trip, 0.43 of a wavefront's cycles in SQ_WAIT_ANY (profiles/r05_wait_counters.txt).  Now the load is a
  // global_load the compiler does not see, issued at the TOP of a trip, and the wait is written out at the trip's END, in
  // front of the loop's register copies (the compiler inserts no wait for a load it does not see: the values must have
  // landed before anything may move them) -- by then the load is a whole trip old.  (Clamped index instead of a branch; the
  // value of a lane without a unit is zeroed behind the wait.)
  auto prefetch = [&](int b, int u) __attribute__((always_inline)) {
    const uint32_t *ptr = desc + max(min(u, nu - 1), 0);
    asm volatile("global_load_dword %0, %1, off" : "=v"(dn[b]) : "v"(ptr) : "memory");
  };
  auto arrive = [&]() __attribute__((always_inline)) {
    if constexpr (UQ == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(dn[0]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(dn[0]), "+v"(dn[UQ - 1]) : : "memory");
  };
  static_assert(UQ <= 2, "arrive() names the first and the last descriptor register");
#pragma unroll
  for (int b = 0; b < UQ; b++) prefetch(b, r + b * g);
  arrive();
#else
#pragma unroll
  for (int b = 0; b < UQ; b++) { const int u = r + b * g; dn[b] = (u < nu) ? desc[u] : 0u; }
#endif
  for (int u0 = r; __any(u0 < nu); u0 += UQ * g) {
    uint32_t d[UQ];
#if MISO_FLAT_ASM_PREFETCH
#pragma unroll
    for (int b = 0; b < UQ; b++) d[b] = (nu > 0 && u0 + b * g < nu) ? dn[b] : 0u;
#pragma unroll
    for (int b = 0; b < UQ; b++) prefetch(b, u0 + (UQ + b) * g);
#else
#pragma unroll
    for (int b = 0; b < UQ; b++) {
      d[b] = dn[b];
      const int un = u0 + (UQ + b) * g;
      dn[b] = (un < nu) ? desc[un] : 0u;   // (measured at K=5: this generic-pointer load under a branch 93.7k events/s; an
                                           //  address-space-1 pointer 87.1k; that plus a clamped, branch-free load 91.1k)
    }
#endif
    uint32_t T[UQ][TW], w[UQ][4];
    // (tried round 4: the chains' whole units first and a trip without word masks when every lane's units are whole -- two
    // instructions per word fewer on paper, no gain measured: profiles/r04_occupancy.txt)
#pragma unroll
    for (int b = 0; b < UQ; b++) {
      const uint32_t *row = reinterpret_cast<const uint32_t *>(thr + ((d[b] >> 4) & 0xFFu) * trow4);
#pragma unroll
      for (int j = 0; j < TW; j++) T[b][j] = row[j];
      const miso_u32x4 u = philox_gibbs<true>(rng, d[b] >> 12, n0r0);
      const int nm = static_cast<int>(~d[b]);   // a word outside the class becomes 0xFFFFFFFF (never below a 32-bit threshold)
#pragma unroll
      for (int x = 0; x < 4; x++) w[b][x] = u.v[x] | static_cast<uint32_t>(__builtin_amdgcn_sbfe(nm, x, 1));
    }
#pragma unroll
    for (int j = 0; j < TW; j++) {
#pragma unroll
      for (int b = 0; b < UQ; b++) count_below(D[j], w[b][0], w[b][1], w[b][2], w[b][3], T[b][j]);
    }
#if MISO_FLAT_ASM_PREFETCH
    arrive();   // the next trip's descriptors: issued at this trip's top
#endif
  }
  int *dl = reinterpret_cast<int *>(wbase + sl + off_dl);
#pragma unroll
  for (int j = 0; j < TW; j++) if (D[j]) atomicAdd(&dl[j], D[j]);
}

}  // namespace

// Workgroups per CU the register budget is set for (runtime.hip sizes the chains per wavefront by the same rule).  Measured
// round 4, 40 000 events (profiles/r04_occupancy.txt): nine to twelve isoforms THREE instead of two (206 -> 168 registers, no
// spills): K = 10 47.7 k -> 49.1 k events/s with five chains per wavefront instead of seven; beyond twelve two.  Up to
// eight isoforms three as before: FOUR (128 registers) gave K = 3 128.5 k -> 135.5 k but K = 4 115.7 k -> 112.4 k (one
// kernel serves both) and nothing at five to eight, where 128 registers spill (94.1 k -> 93.3 k).
#ifndef MISO_FLAT_WGS_4
#define MISO_FLAT_WGS_4 3
#endif
#ifndef MISO_FLAT_WGS_SMALL
#define MISO_FLAT_WGS_SMALL 3
#endif
#ifndef MISO_FLAT_WGS_12
#define MISO_FLAT_WGS_12 3
#endif
#ifndef MISO_FLAT_WGS_LARGE
#define MISO_FLAT_WGS_LARGE 2
#endif
// KS > 0 (round 5): the launch's largest isoform count a.kstride as a compile-time constant -- the slice layout (seventeen
// offsets), the strides and the reciprocals of the flat loops become immediates instead of scalar registers (the kernel spilled
// 227 of them into VGPR lanes and fetched them back with ~500 v_readlane per iteration, an eighth of the scalar step's
// instructions), and the unrolled isoform loops end at KS instead of at the class's KC.  Up to twenty isoforms every count has
// its kernel (kernels_flat_c*.hip); KS = 0: the layout at run time (21 - 32 isoforms).  K = 5 110.0 -> 113.5 k,
// K = 10 58.0 -> 60.7 k events/s with the layout alone (profiles/r05_flat_chunks.txt).
template <int KC, int KS = 0, bool UNI = false>
__global__ __launch_bounds__(256, KC <= 4 ? MISO_FLAT_WGS_4 : (KC <= 8 ? MISO_FLAT_WGS_SMALL : (KC <= 12 ? MISO_FLAT_WGS_12 : MISO_FLAT_WGS_LARGE))) void sampler_flat(const KernelArgs a) {
  static_assert(KS == 0 || (KS <= KC && KS >= 2), "KS: an isoform count of the class KC");
  static_assert(!UNI || KS > 0, "UNI: every event of the launch has KS isoforms");
#define KOF(x) (UNI ? KB : (x))   // a chain's isoform count
  constexpr int KB = KS > 0 ? KS : KC;   // bound of the unrolled isoform loops
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int ks = KS > 0 ? KS : a.kstride;
  const int NC = a.nc, cs = a.cstride, tws = ks - 1, trow = flat_trow(ks);
  const FlatLayout L = flat_layout(ks, cs);
  unsigned char *wbase = smem_flat + static_cast<size_t>(wid) * NC * L.bytes;
  // Which chains this wavefront owns comes from the host (runtime.hip: flat_waves): a.wave_tab[wavefront] = {first
  // chain of the launch's list, chains | FLAT_WIDE}.  Uniform batches: NC chains each; batches whose events differ
  // widely in size (real read counts: 20 ... 10^5 per event, miso.c:845-900 is O(reads) per event): as many
  // consecutive chains as make about the same number of work units per wavefront, and the largest chains one per
  // WORKGROUP (FLAT_WIDE: the four wavefronts each keep the chain's state in their own slice and run the scalar step
  // redundantly -- same inputs, same bits --, the read loop's units are dealt over all 256 lanes, the per-wavefront
  // counts meet through LDS, two barriers per Gibbs step; wavefront 0 writes the outputs).
#ifdef MISO_FLAT_WAVETIME   // tools/archive/wave_time_flat.py: when every chain's wavefront started and how long it ran (100 MHz), diagnostic build
  uint64_t wt_t0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt_t0) : : "memory");
#endif
  const long wave_id = static_cast<long>(blockIdx.x) * 4 + wid;
  const int wt_first = __builtin_amdgcn_readfirstlane(a.wave_tab[2 * wave_id]), wt_n = __builtin_amdgcn_readfirstlane(a.wave_tab[2 * wave_id + 1]);
  const bool wide = (wt_n & FLAT_WIDE) != 0;
  const long first_slot = wt_first;
  const int ncw = wt_n & 0xFF;
  if (ncw == 0) return;   // padding wavefront of the last workgroup (never in a FLAT_WIDE workgroup: no barrier missed)
  const bool writes = !wide || wid == 0;
  const uint32_t k0 = static_cast<uint32_t>(a.seed), k1 = static_cast<uint32_t>(a.seed >> 32);

// (s < 64, the slice < 64 KB: the 24-bit multiply is full rate, the 32-bit one a quarter)
#define FD(s, off) reinterpret_cast<double *>(wbase + __mul24((s), L.bytes) + (off))
#define FI(s, off) reinterpret_cast<int *>(wbase + __mul24((s), L.bytes) + (off))
#define FU(s, off) reinterpret_cast<uint32_t *>(wbase + __mul24((s), L.bytes) + (off))
  // buffer 0 of psi / alpha / lp / tb / lr = the current state and its cached logs, buffer 1 (+ ks) = the proposal
  const int PR = ks;

  // ---- set-up: every chain's constants and class table into its slice ----
  int Kw_rt = 0;
  for (int s = 0; s < ncw; s++) {
    const long slot = first_slot + s;
    const int ev = a.slot_event[slot / a.C];
    const uint32_t chain = static_cast<uint32_t>(slot % a.C);
    const DevEvent E = a.events[ev];
    const int K = E.K;
    Kw_rt = max(Kw_rt, K);
    const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
    const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);
    const uint32_t *gt = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_cls);
    const uint32_t event_id = E.has_id ? E.explicit_id : a.first_event_id + static_cast<uint32_t>(ev);
    for (int k = lane; k < ks; k += 64) {
      const bool in = k < K;
      FD(s, L.cst)[k] = in ? consts[k] : 0.0; FD(s, L.isc)[k] = in ? consts[K + k] : 0.0;
      FD(s, L.hm1)[k] = in ? consts[2 * K + k] : 0.0;
      // miso.c:330-447 START_AUTO: K != 2 -> alpha = 1/(K-1); START_UNIFORM -> 0
      FD(s, L.alpha)[k] = (a.start == MISO_START_AUTO && K != 2 && k < K - 1) ? 1.0 / (K - 1) : 0.0;
      FD(s, L.alpha)[PR + k] = 0.0;
      FD(s, L.psi)[k] = 0.0; FD(s, L.psi)[PR + k] = 0.0;
      FD(s, L.lp)[k] = 0.0; FD(s, L.lp)[PR + k] = 0.0; FD(s, L.tb)[k] = 0.0; FD(s, L.tb)[PR + k] = 0.0;
      FD(s, L.lr)[k] = 0.0; FD(s, L.lr)[PR + k] = 0.0; FD(s, L.tc)[k] = 0.0; FD(s, L.u2)[k] = 0.0;
      FI(s, L.cnt)[k] = 0; FI(s, L.bas)[k] = in ? base[k] : 0; FI(s, L.dl)[k] = 0;
      FU(s, L.ctab)[CLS_WORDS * (cs + 1) + k] = in ? gt[CLS_WORDS * (E.n_dcls + 1) + k] : 0u;   // A_k
    }
    for (int i = lane; i < CLS_WORDS * (E.n_dcls + 1); i += 64) FU(s, L.ctab)[i] = gt[i];
    for (int i = lane; i < cs * trow; i += 64) FU(s, L.thr)[i] = 0u;
    if (lane == 0) {
      int *mi = FI(s, L.misc);
      mi[MI_K] = K; mi[MI_NDRAW] = E.n_draw; mi[MI_NCLS] = E.n_dcls; mi[MI_NUNITS] = E.n_units;
      mi[MI_EVID] = static_cast<int>(event_id); mi[MI_CHAIN] = static_cast<int>(chain);
      mi[MI_ACC] = 1; mi[MI_SLOW] = 0;   // (the first Gibbs step computes every chain's thresholds)
      mi[MI_ACCW] = 0; mi[MI_EV] = ev; mi[MI_NEXT] = -1;
      mi[MI_DESC] = static_cast<int>(static_cast<uint32_t>(E.off_units >> 2));   // dword offset of the event's unit descriptors in the input pool
      mi[MI_LANE0] = 0; mi[MI_LANES] = 0;
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
  Kw_rt = __builtin_amdgcn_readfirstlane(Kw_rt);
  // bound of the isoform loops: the wavefront's largest isoform count -- in a kernel of one isoform count (KS) that count itself:
  // every `k < Kw` is then decided at compile time instead of living in a scalar register pair as a loop-invariant mask (the
  // entries beyond a chain's own K are zeros in its slice and switched off by the chain's `k < lK` as before)
  const int Kw = KS > 0 ? KB : Kw_rt;
  fsync();
  // the wavefront's unit list: chain s owns units [ustart_s, ustart_s + n_units_s); MI_NEXT = the next
  // chain that has units at all
  int total_units = 0;
  {
    int nxt = -1;
    for (int s = ncw - 1; s >= 0; s--) {
      if (lane == 0) FI(s, L.misc)[MI_NEXT] = nxt;
      if (FI(s, L.misc)[MI_NUNITS] > 0) nxt = s;
    }
    for (int s = 0; s < ncw; s++) {
      if (lane == 0) FI(s, L.misc)[MI_USTART] = total_units;
      total_units += FI(s, L.misc)[MI_NUNITS];
    }
  }
  total_units = __builtin_amdgcn_readfirstlane(total_units);
  const int trips = (total_units + 63) / 64;   // units per lane
  // lanes per chain for flat_units_desc: in proportion to the chain's units, every chain with units at least one,
  // 64 in all (largest remainders first; static for the whole run)
  const bool use_desc = a.flat_desc != 0;   // (decided by the host per launch; a second condition here -- the wavefront's largest K --
                                            //  cost 7 % at K=5: 93.5k -> 87.2k events/s, for no visible reason in the read loop itself)
  if (use_desc && total_units > 0 && lane == 0) {
    int sum = 0;
    for (int s = 0; s < ncw; s++) {
      const int nu = FI(s, L.misc)[MI_NUNITS];
      const int gs = nu > 0 ? max(1, static_cast<int>((64L * nu) / total_units)) : 0;
      FI(s, L.misc)[MI_LANES] = gs; sum += gs;
    }
    while (sum != 64) {   // give a lane to the chain with the most units per lane / take one from the one with the fewest
      int best = -1; double bv = 0.0;
      for (int s = 0; s < ncw; s++) {
        const int nu = FI(s, L.misc)[MI_NUNITS], gs = FI(s, L.misc)[MI_LANES];
        if (nu <= 0 || (sum > 64 && gs <= 1)) continue;
        const double v = sum < 64 ? static_cast<double>(nu) / gs : -static_cast<double>(nu) / (gs - 1);
        if (best < 0 || v > bv) { best = s; bv = v; }
      }
      if (best < 0) break;
      FI(best, L.misc)[MI_LANES] += sum < 64 ? 1 : -1;
      sum += sum < 64 ? 1 : -1;
    }
    int l0 = 0;
    for (int s = 0; s < ncw; s++) { FI(s, L.misc)[MI_LANE0] = l0; l0 += FI(s, L.misc)[MI_LANES]; }
  }
  fsync();
  int d_ms = 0, d_r = 0, d_g = 0;   // this lane's chain, rank and group size (flat_units_desc)
  if (use_desc && total_units > 0) {
    for (int s = 0; s < ncw; s++) {
      const int l0 = FI(s, L.misc)[MI_LANE0], gs = FI(s, L.misc)[MI_LANES];
      if (lane >= l0 && lane < l0 + gs) { d_ms = s; d_r = lane - l0; d_g = gs; }
    }
  }
  if (wide) { d_ms = 0; d_r = wid * 64 + lane; d_g = 256; }   // one chain, all four wavefronts' lanes
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
  const int lK = KOF(FI(ls, L.misc)[MI_K]);
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
  // flat loop over (chain s, item j < nper): idx = s * nper + j; lanes past the end shadow item 0 of
  // chain 0 (loads stay inside the slice) and are switched off by `on`
#define FLAT_BEGIN(nper, inv)                                                   \
  for (int base_ = 0; base_ < ncw * (nper); base_ += 64) {                      \
    const int idx_ = base_ + lane;                                              \
    const bool on = idx_ < ncw * (nper);                                        \
    const int s = on ? static_cast<int>((static_cast<float>(idx_) + 0.5f) * (inv)) : 0; \
    const int j = on ? idx_ - s * (nper) : 0;
#define FLAT_END }
  // a leader's vector: all loads first (one LDS latency), then the reference's left-to-right arithmetic
  // (in chunks of CH isoforms, so that the K <= 32 kernel does not hold whole vectors in registers)
  // Round 5: chunks of FOUR up to twelve isoforms.  The joint score below holds six vectors of CH doubles at once; with CH = 8
  // that is where the K <= 8 kernel needs 209 registers, and at the 168 of three workgroups per CU 44 loop-invariant values sat
  // in scratch (176 bytes per lane, ~30 reloads per iteration, most of the row's HBM traffic).  With four: 183 / 48 bytes /
  // 5 reloads; K = 5 / 6 / 7 / 8 104.3 -> 109.3 / 95.2 -> 100.0 / 90.0 -> 93.8 / 70.3 -> 72.7 k events/s, K = 10 / 12 55.0 -> 57.7 /
  // 40.8 -> 42.2 k (192 -> 128 bytes); 13 - 16 isoforms (two workgroups per CU, no scratch either way) keep eight: 29.76 vs
  // 29.52 k (profiles/r05_flat_chunks.txt).  Same sums in the same order: the chunk is how many operands are loaded at once.
#ifndef MISO_FLAT_CH_SMALL
#define MISO_FLAT_CH_SMALL 4   // KC <= 12
#endif
#ifndef MISO_FLAT_CH_MID
#define MISO_FLAT_CH_MID 8     // KC 16
#endif
  constexpr int CH = KC <= 12 ? (MISO_FLAT_CH_SMALL < KB ? MISO_FLAT_CH_SMALL : KB) : (KC <= 16 ? MISO_FLAT_CH_MID : 4);
#define CHUNKS_BEGIN                                                            \
  _Pragma("unroll") for (int k0 = 0; k0 < KB; k0 += CH) {                       \
    if (k0 < Kw) {
#define CHUNKS_END }}
#define LOADC(v, p)                                                             \
  double v[CH];                                                                 \
  _Pragma("unroll") for (int i_ = 0; i_ < CH; i_++) v[i_] = (k0 + i_ < Kw) ? (p)[k0 + i_] : 0.0;
#define EACH(k) _Pragma("unroll") for (int i_ = 0, k = k0; i_ < CH; i_++, k++) if (k < Kw)

#ifdef MISO_K2_PROFILE
  uint64_t fp_mh = 0, fp_thr = 0, fp_loop = 0;
#endif

  // ---- alpha' = alpha + sd z ; psi' = logit_inv(alpha') (miso.c:449-471), then the psi-only parts of
  // both scores of the new point: lp = log x, tb = lp + cst, lr = log(x_k / x_K'), jacobian.
  // SRC / DST: buffer offsets (0 = current, PR = proposal) of alpha read / everything written. ----
  auto propose_and_logs = [&](uint32_t iter, int SRC, int DST, double &jac_out) __attribute__((always_inline)) {
    FLAT_BEGIN(tws, inv_k1)   // pass 1 (qnorm) + pass 2 (exp), one normal per lane
      const int *mi = FI(s, L.misc);
      const int K = KOF(mi[MI_K]);
      const uint32_t evid = static_cast<uint32_t>(mi[MI_EVID]), chain = static_cast<uint32_t>(mi[MI_CHAIN]);
      const double al = FD(s, L.alpha)[SRC + j], sd = FD(s, L.sx)[SX_SD];
      const int w = 2 + 2 * j;
      const miso_u32x4 b = miso_draw_block(a.seed, evid, chain, iter, MISO_SITE_MH, static_cast<uint32_t>(w >> 2));
      const bool odd = (j & 1) != 0;   // w & 3 = 0 for odd j, 2 for even j
      const double z = miso_det_norm_from_unif(miso_u01(odd ? b.v[0] : b.v[2]), miso_u01(odd ? b.v[1] : b.v[3]));
      const double an = al + sd * z;
      const double ex = miso_det_exp(an);
      if (on && j < K - 1) {
        FD(s, L.alpha)[DST + j] = an;
        FD(s, L.tc)[j] = ex;
        if (j == 0) FI(s, L.misc)[MI_ACCW] = static_cast<int>(b.v[0]);   // block 0, word 0 (miso.c:870)
      }
    FLAT_END
    fsync();
    if (leader) {
      double acc = 0.0;
      CHUNKS_BEGIN
        LOADC(tc, FD(ls, L.tc))
        EACH(k) acc = (k < lK - 1) ? acc + tc[i_] : acc;
      CHUNKS_END
      FD(ls, L.sx)[SX_SUMEXP] = acc + 1.0;
    }
    fsync();
    FLAT_BEGIN(tws, inv_k1)
      const int K = KOF(FI(s, L.misc)[MI_K]);
      const double q = FD(s, L.tc)[j] / FD(s, L.sx)[SX_SUMEXP];
      if (on && j < K - 1) FD(s, L.psi)[DST + j] = q;
    FLAT_END
    fsync();
    if (leader) {
      double *x = FD(ls, L.psi) + DST;
      double sumpsi = 0.0, ltheta = 1.0, prod = 1.0;
      CHUNKS_BEGIN
        LOADC(xv, x)
        EACH(k) {
          const bool in = k < lK - 1;
          sumpsi = in ? sumpsi + xv[i_] : sumpsi; ltheta = in ? ltheta - xv[i_] : ltheta; prod = in ? prod * xv[i_] : prod;
        }
      CHUNKS_END
      x[lK - 1] = 1 - sumpsi;
      FD(ls, L.sx)[SX_LTHETA] = ltheta;
      jac_out = 1.0 / prod / ltheta;
    }
    fsync();
    FLAT_BEGIN(2 * ks - 1, inv_2k)   // pass 3 (log): 2K - 1 arguments per chain
      const int K = KOF(FI(s, L.misc)[MI_K]);
      const bool firsthalf = j < ks;
      const int k = firsthalf ? j : j - ks;
      const double xv = FD(s, L.psi)[DST + k], lt = FD(s, L.sx)[SX_LTHETA], cst = FD(s, L.cst)[k];
      const double r = miso_det_log(firsthalf ? xv : xv / lt);
      // (iterations) the lane that has the proposal's log ratio also forms the Gaussian parts of the two
      // proposal densities (miso.c:110-117): proposal -> current uses the current psi's log ratios against
      // alpha', current -> proposal the proposal's against alpha
      const double sigma = FD(s, L.sx)[SX_SIGMA];
      const double t1 = FD(s, L.lr)[k] - FD(s, L.alpha)[PR + k];
      const double t2 = r - FD(s, L.alpha)[k];
      const double g1 = (-0.5) * t1 * t1 / sigma, g2 = (-0.5) * t2 * t2 / sigma;
      if (on && (firsthalf ? (k < K) : (k < K - 1))) {
        if (firsthalf) { FD(s, L.lp)[DST + k] = r; FD(s, L.tb)[DST + k] = r + cst; }
        else {
          FD(s, L.lr)[DST + k] = r;
          if (DST != 0) { FD(s, L.tc)[k] = g1; FD(s, L.u2)[k] = g2; }
        }
      }
    FLAT_END
    fsync();
  };
  auto leader_max = [&](int BUF) __attribute__((always_inline)) {   // miso.c:137-140: maxv starts at entry 0
    double maxv = 0.0;
    if (leader) {
      maxv = (FD(ls, L.tb) + BUF)[0];
      CHUNKS_BEGIN
        LOADC(tb, FD(ls, L.tb) + BUF)
        EACH(k) maxv = (k >= 1 && k < lK && tb[i_] > maxv) ? tb[i_] : maxv;
      CHUNKS_END
      FD(ls, L.sx)[SX_MAXV] = maxv;
    }
    return maxv;
  };
  auto count_of = [&](int s, int k) __attribute__((always_inline)) { return FI(s, L.bas)[k] + FI(s, L.cnt)[k]; };

  // ---- per-read picks by direct evaluation of the reference's scan (miso.c:11-22, 69-80): the final
  // assignment of chain 0 (miso.c:943-946) and the fallback when a threshold does not fit 32 bits ----
  // (always_inline, all of them: outlined -- as the K = 10 kernel of one isoform count did with direct_chain -- a lambda reaches the
  // kernel's locals through its capture block in scratch and the slices through generic pointers: 541 -> 733 ms)
  auto direct_chain = [&](int s, uint32_t iter, bool count, bool write) __attribute__((always_inline)) {
    const int *mi = FI(s, L.misc);
    const int K = KOF(mi[MI_K]), n_draw = mi[MI_NDRAW];
    const DevEvent E = a.events[mi[MI_EV]];
    const uint32_t *masks = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_draw);
    uint8_t *drawass = a.out_pool + E.off_drawass;
    const double *psi = FD(s, L.psi);
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
  auto gibbs = [&](uint32_t iter) __attribute__((always_inline)) {
    FPROF_T(t0);
    // thresholds: one lane per (chain, class) -- of the chains whose psi CHANGED.  The thresholds are a function of psi
    // alone, and the Metropolis-Hastings step keeps psi in 45 % (ten isoforms) to 65 % (three to five) of the iterations
    // (measured acceptance rates 0.32 - 0.47 / 0.55 / 0.68 at K = 3 - 5 / 10 / 16): a chain that rejected its proposal keeps
    // its rows -- and its MI_SLOW flag -- from the step before.  The chains that accepted are numbered through a bit mask.
    uint64_t accmask = 0;
    // (the host turns it on where the thresholds weigh enough -- from six isoforms on and for small events, a.flat_thr_skip,
    // runtime.hip launch_flat; measured: + 4.5 ... 6.4 % there, - 1 ... 2 % at three to five isoforms x 1000 reads)
    for (int s = 0; s < ncw; s++) accmask |= static_cast<uint64_t>(!a.flat_thr_skip || FI(s, L.misc)[MI_ACC] != 0) << s;
    const int n_acc = __popcll(accmask);
#ifdef MISO_FLAT_SKIP_THR
    if (a.M < 0)
#endif
    for (int base_ = 0; base_ < n_acc * cs; base_ += 64) {
      const int idx_ = base_ + lane;
      const bool on = idx_ < n_acc * cs;
      const int ai = on ? static_cast<int>((static_cast<float>(idx_) + 0.5f) * inv_cs) : 0;
      const int j = on ? idx_ - ai * cs : 0;
      int s = on ? ai : 0;
      if (a.flat_thr_skip) {   // the ai-th chain that accepted
        uint64_t rest = accmask;
        for (int t = 0; t < ai; t++) rest &= rest - 1;
        s = on ? static_cast<int>(__builtin_ctzll(rest | (1ull << 63))) : 0;
      }
      const int *mi = FI(s, L.misc);
      const int K = KOF(mi[MI_K]), ncls = mi[MI_NCLS];
      const uint32_t m = FU(s, L.ctab)[CLS_WORDS * j];
      const double *psi = FD(s, L.psi);
      double ps[KB];
#pragma unroll
      for (int k = 0; k < KB; k++) ps[k] = (k < Kw) ? psi[k] : 0.0;
      const bool mine = on && j < ncls;
      // total weight, ascending isoforms (miso.c:11-22); +0.0 for the others leaves the bits alone
      double T = 0.0;
#pragma unroll
      for (int k = 0; k < KB; k++) if (k < Kw) T = T + ((k < K && ((m >> k) & 1u)) ? ps[k] : 0.0);
      const double inv = 4294967296.0 / T;
      const bool tnormal = T >= 1e-280 && T <= 1e280;   // products with u 2^-32 stay normal
      const bool le = __popc(m) != 2;
      const int kmax = 31 - __clz(static_cast<int>(m));
      uint32_t *th = FU(s, L.thr) + j * trow;
      double cum = 0.0;
      uint32_t run = 0u;
      bool slow = false;
#pragma unroll
      for (int k = 0; k < KB - 1; k++) {
        if (k < Kw - 1) {
          const bool member = k < K && ((m >> k) & 1u);
          cum = cum + (member ? ps[k] : 0.0);
          // a member before the last one: first j with u < t_j == u < max(t_0..t_j)
          const bool use = mine && member && k < kmax;
          const double est = cum * inv;
          double t;
          if (__any(use && !(tnormal && est >= 2.0 && est <= 4294967293.0))) t = flat_threshold(le, cum, T, est);
          else t = flat_threshold_fast(le, cum, T, est);
          slow |= use && t >= 4294967296.0;
          const uint32_t tu = static_cast<uint32_t>(t);
          run = (use && tu > run) ? tu : run;
          if (mine && k < K - 1) th[k] = (k < kmax) ? run : 0u;
        }
      }
      if (slow) FI(s, L.misc)[MI_SLOW] = 1;   // (cleared by the chain's leader when it accepted; every writer writes 1)
    }
    fsync();
    bool slow = false;
    for (int s = 0; s < ncw; s++) slow |= FI(s, L.misc)[MI_SLOW] != 0;
    fsync();   // (dl is zero here: set-up, and the end of every Gibbs step)
    FPROF_T(t1);
    FPROF_ADD(fp_thr, t0, t1);
    if (__any(slow)) {   // a non-final threshold of 2^32 cannot be held in 32 bits: direct path this time
      FLAT_BEGIN(ks, inv_k)
        if (on) FI(s, L.cnt)[j] = 0;
      FLAT_END
      fsync();
      for (int s = 0; s < ncw; s++) direct_chain(s, iter, true, false);
      fsync();
      return;
    }
    const int tww = Kw_rt - 1;
#ifdef MISO_FLAT_SKIP_LOOP   // instruction accounting (tools/flat_phase_valu.sh): results are wrong
    if (a.M >= 0) return;
#endif
    FlatUnitsArgs ua;
    ua.woff = wid * NC * L.bytes; ua.slice = L.bytes; ua.off_ctab = L.ctab; ua.off_thr = L.thr; ua.off_misc = L.misc; ua.off_dl = L.dl;
    ua.trow = trow; ua.trips = trips; ua.iter = iter; ua.k0 = k0; ua.k1 = k1;
#define MISO_FUNITS(TW) { if (use_desc) flat_units_desc<TW>(ua, reinterpret_cast<const uint32_t *>(a.in_pool), d_ms, d_r, d_g); else flat_units<TW>(ua, s0, c0, i0, n_mine); }
    // (tww <= KB - 1: a kernel of one isoform count carries the one read loop it can reach)
    if constexpr (KC == 4) { if (KB <= 3 || tww <= 2) MISO_FUNITS(2) else MISO_FUNITS(3) }
    else if constexpr (KC == 8) { if (KB <= 5 || tww <= 4) MISO_FUNITS(4) else if (KB <= 6 || tww == 5) MISO_FUNITS(5) else if (KB <= 7 || tww == 6) MISO_FUNITS(6) else MISO_FUNITS(7) }
    else if constexpr (KC == 12) { if (KB <= 10 || tww <= 9) MISO_FUNITS(9) else MISO_FUNITS(11) }
    else if constexpr (KC == 16) { MISO_FUNITS(15) }
    else { if (KB <= 20 || tww <= 19) MISO_FUNITS(19) else if (tww <= 23) MISO_FUNITS(23) else MISO_FUNITS(31) }
#undef MISO_FUNITS
    fsync();
    if (wide) {
      // every wavefront's D_k -> the sum of all four, in every wavefront's own slice (the slices start NC slices apart)
      __syncthreads();
      int tot = 0;
      if (lane <= trow) {
#pragma unroll
        for (int w = 0; w < 4; w++) tot += reinterpret_cast<const int *>(smem_flat + static_cast<size_t>(w) * NC * L.bytes + L.dl)[lane];
      }
      __syncthreads();
      if (lane <= trow) FI(0, L.dl)[lane] = tot;
      fsync();
    }
    // D_k (+ the reads of classes that end at or before k) -> picks per isoform
    FLAT_BEGIN(ks, inv_k)
      const int *mi = FI(s, L.misc);
      const int K = KOF(mi[MI_K]), nd = mi[MI_NDRAW];
      const uint32_t *A = FU(s, L.ctab) + CLS_WORDS * (cs + 1);
      const int *dl = FI(s, L.dl);
      const int jm = max(j - 1, 0);
      const int dj = dl[j], dm = dl[jm], aj = static_cast<int>(A[j]), am = static_cast<int>(A[jm]);
      const int hi = (j < K - 1) ? dj + aj : nd;
      const int lo = (j > 0) ? dm + am : 0;
      if (on && j < K) FI(s, L.cnt)[j] = hi - lo;
    FLAT_END
    fsync();
    FLAT_BEGIN(ks, inv_k)   // D_k back to zero for the next step
      if (on) { FI(s, L.dl)[j] = 0; if (j == 0) for (int x = ks; x <= trow; x++) FI(s, L.dl)[x] = 0; }
    FLAT_END
    FPROF_T(t2);
    FPROF_ADD(fp_loop, t1, t2);
  };

  // ---- initial state: miso.c:834 (alpha + sd z in place), cached logs, log-sum-exp, miso.c:841 ----
  propose_and_logs(MISO_ITER_INIT, 0, 0, l_jac);
  {
    const double maxv = leader_max(0);
    fsync();
    FLAT_BEGIN(ks, inv_k)
      const int K = KOF(FI(s, L.misc)[MI_K]);
      const double r = miso_det_exp(FD(s, L.tb)[j] - FD(s, L.sx)[SX_MAXV]);
      if (on && j < K) FD(s, L.tc)[j] = r;
    FLAT_END
    fsync();
    if (leader) {
      double acc = 0.0;
      CHUNKS_BEGIN
        LOADC(tc, FD(ls, L.tc))
        EACH(k) acc = (k < lK) ? acc + tc[i_] : acc;
      CHUNKS_END
      l_lse = miso_det_log(acc) + maxv;
    }
    fsync();
  }
  gibbs(MISO_ITER_INIT);

  const bool tracing = __any(LE_.off_trace != NO_TRACE);   // all events of a batch trace or none
  RoundOpen ro(a);
  for (int m = 0; m < a.M; m++) {
    const bool opens = ro.at(a, m);   // a round's first iteration: no proposal terms (miso.c:866)
    prio_by_progress(a, m);
    // this iteration's view of the counts: the leader's registers (hash, both joint scores)
    int cn[KB];
    if (leader) {
      const int *bas = FI(ls, L.bas), *cnt = FI(ls, L.cnt);
#pragma unroll
      for (int k = 0; k < KB; k++) cn[k] = (k < Kw) ? bas[k] + cnt[k] : 0;
#pragma unroll
      for (int k = 0; k < KB; k++)
        if (k < Kw) hash = (k < lK) ? (hash ^ static_cast<uint32_t>(cn[k])) * 0x100000001B3ull : hash;
    }
    if (tracing) {
      FLAT_BEGIN(ks, inv_k)
        const int *mi = FI(s, L.misc);
        const int K = KOF(mi[MI_K]);
        if (on && j < K) {
          const uint64_t to = (static_cast<uint64_t>(static_cast<uint32_t>(mi[MI_TRACE_HI])) << 32) | static_cast<uint32_t>(mi[MI_TRACE_LO]);
          if (writes) reinterpret_cast<int32_t *>(a.out_pool + to)[(static_cast<size_t>(m) * a.C + mi[MI_CHAIN]) * K + j] = count_of(s, j);
        }
      FLAT_END
    }
    FPROF_T(m0);
    double jacN = 0.0;
    propose_and_logs(static_cast<uint32_t>(m), 0, PR, jacN);                          // passes 1, 2, 3
    const double maxN = leader_max(PR);
    if (leader) {
      double e1 = 0.0, e2 = 0.0;
      CHUNKS_BEGIN
        LOADC(tc, FD(ls, L.tc))
        LOADC(u2, FD(ls, L.u2))
        EACH(k) { e1 = (k < lK - 1) ? e1 + tc[i_] : e1; e2 = (k < lK - 1) ? e2 + u2[i_] : e2; }
      CHUNKS_END
      FD(ls, L.sx)[SX_E1] = e1; FD(ls, L.sx)[SX_E2] = e2;
    }
    fsync();
    FLAT_BEGIN(ks + 2, inv_k2)                                                         // pass 4: exp
      const int K = KOF(FI(s, L.misc)[MI_K]);
      const bool iso = j < ks;
      const double *sx = FD(s, L.sx);
      const double tbv = FD(s, L.tb)[PR + (iso ? j : 0)], mx = sx[SX_MAXV], e1 = sx[SX_E1], e2 = sx[SX_E2];
      const double r = miso_det_exp(iso ? tbv - mx : (j == ks ? e1 : e2));
      if (on && (iso ? (j < K) : true)) {
        if (iso) FD(s, L.tc)[j] = r; else FD(s, L.sx)[SX_X1 + (j - ks)] = r;
      }
    FLAT_END
    fsync();
    if (leader) {
      double *sx = FD(ls, L.sx);
      const double x1 = sx[SX_X1], x2 = sx[SX_X2];
      double sumtc = 0.0;
      CHUNKS_BEGIN
        LOADC(tc, FD(ls, L.tc))
        EACH(k) sumtc = (k < lK) ? sumtc + tc[i_] : sumtc;
      CHUNKS_END
      sx[SX_LA0] = sumtc; sx[SX_LA1] = l_covar * l_jac * x1; sx[SX_LA2] = l_covar * jacN * x2;
    }
    fsync();
    for (int base_ = 0; base_ < ncw * 3; base_ += 64) {                                // pass 5: log
      const int idx_ = base_ + lane;
      const bool on = idx_ < ncw * 3;
      const int s = on ? idx_ / 3 : 0, j = on ? idx_ - 3 * s : 0;
      const double r = miso_det_log(FD(s, L.sx)[SX_LA0 + j]);
      if (on) FD(s, L.sx)[SX_LR0 + j] = r;
    }
    fsync();
    double cJS = 0.0;
    if (leader) {
      // joint log score of the proposal and of the current point for the current counts (miso.c:243-307)
      const double *sx = FD(ls, L.sx);
      const double lseN = sx[SX_LR0] + maxN, ptoCS = sx[SX_LR1], ctoPS = sx[SX_LR2];
      const uint32_t accw = static_cast<uint32_t>(FI(ls, L.misc)[MI_ACCW]);
      // both scores in one walk: rp / ap = the two count-weighted sums, pq = the Dirichlet part; [0] proposal, [1] current
      double rp[2] = {0.0, 0.0}, ap[2] = {0.0, 0.0}, pq[2] = {0.0, 0.0};
      const double lse2[2] = {lseN, l_lse};
      CHUNKS_BEGIN
        LOADC(isc, FD(ls, L.isc))
        LOADC(hm1, FD(ls, L.hm1))
        LOADC(lpN, FD(ls, L.lp) + PR)
        LOADC(tbN, FD(ls, L.tb) + PR)
        LOADC(lpC, FD(ls, L.lp))
        LOADC(tbC, FD(ls, L.tb))
        EACH(k) {
          const bool in = k < lK, nz = in && cn[k] != 0;
          const double ck = static_cast<double>(cn[k]);
          rp[0] = nz ? rp[0] + ck * isc[i_] : rp[0];
          ap[0] = nz ? ap[0] + ck * (tbN[i_] - lse2[0]) : ap[0];
          pq[0] = in ? pq[0] + hm1[i_] * lpN[i_] : pq[0];
          rp[1] = nz ? rp[1] + ck * isc[i_] : rp[1];
          ap[1] = nz ? ap[1] + ck * (tbC[i_] - lse2[1]) : ap[1];
          pq[1] = in ? pq[1] + hm1[i_] * lpC[i_] : pq[1];
        }
      CHUNKS_END
      double pj[2];
#pragma unroll
      for (int which = 0; which < 2; which++) {
        double psiProb = pq[which];
        psiProb = psiProb + l_lg_sum;
        psiProb = psiProb - l_lg_each;
        pj[which] = rp[which] + ap[which] + psiProb;
      }
      const double pp = pj[0], pc = pj[1];
      const double acceptP = !opens ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);  // pass 6
      const bool acc = (acceptP >= 1) || (miso_u01(accw) < acceptP);
      cJS = pc;
      FI(ls, L.misc)[MI_ACC] = acc ? 1 : 0;
      if (acc) FI(ls, L.misc)[MI_SLOW] = 0;   // new psi: the next Gibbs step recomputes this chain's thresholds
      if (acc) { l_jac = jacN; l_lse = lseN; cJS = pp; accepted++; }
    }
    fsync();
    // accepted: the proposal and its cached logs become the current state; the same lanes record the
    // sample when one is due (miso.c:882-893)
    const bool rec = m >= a.B && lagCounter == a.lag - 1;
    FLAT_BEGIN(ks, inv_k)
      const int *mi = FI(s, L.misc);
      const int acc = mi[MI_ACC], K = KOF(mi[MI_K]);
      const double v0 = FD(s, L.psi)[PR + j], v1 = FD(s, L.alpha)[PR + j], v2 = FD(s, L.lp)[PR + j],
                   v3 = FD(s, L.tb)[PR + j], v4 = FD(s, L.lr)[PR + j];
      const double c0 = FD(s, L.psi)[j];
      if (on && acc) {
        FD(s, L.psi)[j] = v0; FD(s, L.alpha)[j] = v1; FD(s, L.lp)[j] = v2; FD(s, L.tb)[j] = v3; FD(s, L.lr)[j] = v4;
      }
#ifndef MISO_FLAT_STORE_AFTER
#define MISO_FLAT_STORE_AFTER 0   // 1: the iteration's sample leaves after the Gibbs step's loads (a pass of its own); measured, round 5: K = 5 410.4 vs 415.1 ms, K = 10 743.1 vs 748.0 ms (profiles/r05_store_after.txt); 0: from this pass
#endif
      if (!MISO_FLAT_STORE_AFTER && rec && on && j < K && writes) {
        const uint64_t so = (static_cast<uint64_t>(static_cast<uint32_t>(mi[MI_SAMP_HI])) << 32) | static_cast<uint32_t>(mi[MI_SAMP_LO]);
        const size_t col = static_cast<size_t>(noS) + mi[MI_CHAIN];
        reinterpret_cast<double *>(a.out_pool + so)[col * K + j] = acc ? v0 : c0;
      }
    FLAT_END
    fsync();
    FPROF_T(m1);
    FPROF_ADD(fp_mh, m0, m1);
    const int rec_col = noS;
    if (m >= a.B) {
      if (rec) {
        if (!MISO_FLAT_STORE_AFTER && leader && writes) reinterpret_cast<double *>(a.out_pool + LE_.off_loglik)[static_cast<size_t>(noS) + lchain] = cJS;
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    gibbs(static_cast<uint32_t>(m));
    // the sample BEHIND the Gibbs step's loads (vmcnt counts loads and stores in order: the step's first descriptor load
    // used to wait for the sample stored just before it to be acknowledged); psi is not touched by the step
    if (MISO_FLAT_STORE_AFTER && rec) {
      FLAT_BEGIN(ks, inv_k)
        const int *mi = FI(s, L.misc);
        const int K = KOF(mi[MI_K]);
        if (on && j < K && writes) {
          const uint64_t so = (static_cast<uint64_t>(static_cast<uint32_t>(mi[MI_SAMP_HI])) << 32) | static_cast<uint32_t>(mi[MI_SAMP_LO]);
          const size_t col = static_cast<size_t>(rec_col) + mi[MI_CHAIN];
          reinterpret_cast<double *>(a.out_pool + so)[col * K + j] = FD(s, L.psi)[j];
        }
      FLAT_END
      if (leader && writes) reinterpret_cast<double *>(a.out_pool + LE_.off_loglik)[static_cast<size_t>(rec_col) + lchain] = cJS;
    }
  }
  if (leader)
    for (int k = 0; k < Kw; k++)
      if (k < lK) hash = (hash ^ static_cast<uint32_t>(count_of(ls, k))) * 0x100000001B3ull;
  if (tracing) {
    FLAT_BEGIN(ks, inv_k)
      const int *mi = FI(s, L.misc);
      const int K = KOF(mi[MI_K]);
      if (on && j < K) {
        const uint64_t to = (static_cast<uint64_t>(static_cast<uint32_t>(mi[MI_TRACE_HI])) << 32) | static_cast<uint32_t>(mi[MI_TRACE_LO]);
        if (writes) reinterpret_cast<int32_t *>(a.out_pool + to)[(static_cast<size_t>(a.M) * a.C + mi[MI_CHAIN]) * K + j] = count_of(s, j);
      }
    FLAT_END
  }
  // chain 0's final picks, read by read (miso.c:943-946): the last Gibbs step's draws once more
  for (int s = 0; s < ncw; s++)
    if (FI(s, L.misc)[MI_CHAIN] == 0 && writes)
      direct_chain(s, a.M > 0 ? static_cast<uint32_t>(a.M - 1) : MISO_ITER_INIT, false, true);
#ifdef MISO_K2_PROFILE
  if (leader && lchain == 0 && a.M > 8) {
    double *loglik = reinterpret_cast<double *>(a.out_pool + LE_.off_loglik);
    loglik[0] = static_cast<double>(fp_mh); loglik[1] = static_cast<double>(fp_thr); loglik[2] = static_cast<double>(fp_loop);
  }
#endif
  if (leader && writes) {
    ChainStats *st = reinterpret_cast<ChainStats *>(a.out_pool + LE_.off_stats) + lchain;
    st->counts_hash = hash; st->accepted = accepted;
    st->hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
#ifdef MISO_FLAT_WAVETIME
    uint64_t wt_t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt_t1) : : "memory");
    st->hw_id = static_cast<uint32_t>(wt_t1 - wt_t0);
    st->counts_hash = wt_t0;   // when it started
#endif
  }
#undef KOF
#undef FD
#undef FI
#undef FU
#undef FLAT_BEGIN
#undef FLAT_END
#undef LOADC
#undef EACH
#undef CHUNKS_BEGIN
#undef CHUNKS_END
}

}  // namespace miso

// kernels_k2.inl -- the two-isoform single-end sampler (SE / RI / A3SS / A5SS / MXE events:
// BASELINE.json configs[1]), lane-packed for CDNA4.
//
// Why a second kernel: with K = 2 the per-iteration scalar math (propose + Metropolis-Hastings,
// miso.c:449-552: ~11 f64 transcendentals and ~13 f64 divisions) costs as much as the whole Gibbs
// sweep over ~500 ambiguous reads, and both are VALU-issue bound on gfx950 (f64 = 4 cycles per
// wave instruction, measured).  One wavefront per chain (sampler_wave) wastes 63/64 of the scalar
// issue slots.  Here a chain owns G lanes (G = 1..64, a power of two picked by the host):
//
//   * MH step, G >= 4: the FOUR lanes of every quad evaluate ONE transcendental routine on FOUR
//     different arguments (log psi'_0 | log psi'_1 | logit psi'_0, then three exps, then three
//     logs) and exchange results with DPP quad broadcasts: 5 routine calls per iteration instead
//     of 11.  Every value is still produced by the same miso_detmath routine on the same input,
//     so the bits are unchanged.  Terms that depend only on the CURRENT psi are cached and
//     swapped on acceptance.
//   * Gibbs step: the G lanes stride over the chain's draw quads (one Philox4x32 block = the
//     uniforms of four consecutive ambiguous reads, two blocks in flight per trip).  With two
//     compatible isoforms the reference's test  U * (psi0 + psi1) < psi0  (miso.c:69-73) is
//     monotone in the 32-bit uniform, so it becomes ONE u32 compare against the threshold
//     t = #{u : fl(fl(u 2^-32)(psi0+psi1)) < psi0}, found exactly once per iteration.  Round 0 of
//     Philox is partly hoisted: the counter words (iteration, site|chain, event) are chain
//     constants.
//   * a log2(G)-step cross-lane add gives the chain its count.
// No LDS, no barriers; chain state lives in VGPRs.  Reads with fewer than two compatible isoforms
// never reach the device (host.hpp PackedEvent).
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "miso_amd.h"
#include "miso_detmath.h"
#include "miso_philox.h"
#include "gibbs_rng.hpp"
#include "coop.hpp"
#include "miso_binomial.h"
#include "detmath_n.hpp"

#pragma clang fp contract(off)

#ifndef MISO_K2_TAB_REGS
#define MISO_K2_TAB_REGS 0   // 1: the Metropolis-Hastings step's exp / log coefficients in VGPRs instead of scalar loads at every call
#endif
#ifndef MISO_K2_UQ
#define MISO_K2_UQ 2   // Philox blocks in flight per lane in the single-end read loop
#endif

#ifdef MISO_K2_PROFILE
#define PROF_T(var) const uint64_t var = __builtin_readcyclecounter()
#define PROF_ADD(acc, t0, t1) acc += (t1) - (t0)
#else
#define PROF_T(var)
#define PROF_ADD(acc, t0, t1)
#endif

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

struct K2Consts {
  double cst0, cst1, is0, is1, hm0, hm1, lg_sum, lg_each, sigma, sd, covar;
};

// scalar form (set-up, and the whole MH step when G < 4)
__device__ __forceinline__ PsiTerms psi_terms(double x0, double x1, double cst0, double cst1) {
  PsiTerms t;
  t.x0 = x0; t.x1 = x1;
  t.lx0 = miso_det_log(x0);
  t.lx1 = miso_det_log(x1);
  const double lp0 = t.lx0 + cst0, lp1 = t.lx1 + cst1;
  const bool m1 = lp1 > lp0;  // miso.c:137-140: maxv starts at entry 0
  const double maxv = m1 ? lp1 : lp0;
  const double ex0 = miso_det_exp(lp0 - maxv), ex1 = miso_det_exp(lp1 - maxv);
  const double lse = miso_det_log((0.0 + ex0) + ex1) + maxv;
  t.lpn0 = lp0 - lse;
  t.lpn1 = lp1 - lse;
  const double ltheta = 1.0 - x0;
  t.lgt = miso_det_log(x0 / ltheta);
  t.pr = 1.0 / (1.0 * x0) / ltheta;
  return t;
}

// miso.c:243-307 with the per-read sums taken from the counts
template <bool PE>
__device__ __forceinline__ double joint(const PsiTerms &t, int c0, int c1, const K2Consts &c, double readProbPE) {
  double readProb = 0.0, assProb = 0.0, psiProb = 0.0;
  if (c0 != 0) { if (!PE) readProb = readProb + static_cast<double>(c0) * c.is0; assProb = assProb + static_cast<double>(c0) * t.lpn0; }
  if (c1 != 0) { if (!PE) readProb = readProb + static_cast<double>(c1) * c.is1; assProb = assProb + static_cast<double>(c1) * t.lpn1; }
  if (PE) readProb = readProbPE;  // miso_paired.c:157-163, summed in 2^-26 fixed point
  psiProb = psiProb + c.hm0 * t.lx0;
  psiProb = psiProb + c.hm1 * t.lx1;
  psiProb = psiProb + c.lg_sum;
  psiProb = psiProb - c.lg_each;
  return readProb + assProb + psiProb;
}

// miso.c:97-122 for len = 1: exponent of the logistic-normal density of theta around mu
__device__ __forceinline__ double prop_exponent(double lgt, double mu, double sigma) {
  const double tmp = lgt - mu;
  return 0.0 + (-0.5) * tmp * tmp / sigma;
}

// #{u in [0, 2^32) : fl(fl(u * 2^-32) * T) < p0}  -- the reference's two-way draw as a threshold
__device__ __forceinline__ bool k2_pred(int64_t u, double p0, double T) {
  // the reference's test for uniform word u; u outside [0, 2^32) extends it monotonically
  if (u < 0) return true;
  if (u >= 4294967296ll) return false;
  return static_cast<double>(static_cast<uint32_t>(u)) * (1.0 / 4294967296.0) * T < p0;
}

__device__ __forceinline__ uint64_t k2_threshold_exact(double p0, double T) {
  double est = p0 / T * 4294967296.0;
  est = (est > 0.0) ? est : 0.0;  // also catches NaN
  est = (est > 4294967296.0) ? 4294967296.0 : est;
  const int64_t t0 = static_cast<int64_t>(est);
  // The predicate is monotone in u and est is within one step of the boundary, so the threshold is
  // t0 - 1 + (number of true tests among t0-1, t0, t0+1); the two outer tests guard that claim.
  const int n = k2_pred(t0 - 1, p0, T) + k2_pred(t0, p0, T) + k2_pred(t0 + 1, p0, T);
  int64_t t = t0 - 1 + n;
  if (!k2_pred(t0 - 2, p0, T) || k2_pred(t0 + 2, p0, T)) {  // never taken for finite psi; exact fallback
    t = t0 < 0 ? 0 : (t0 > 4294967296ll ? 4294967296ll : t0);
    for (int g = 0; g < 4096 && t > 0 && !k2_pred(t - 1, p0, T); g++) t--;
    for (int g = 0; g < 4096 && t < 4294967296ll && k2_pred(t, p0, T); g++) t++;
  }
  t = t < 0 ? 0 : t;
  return static_cast<uint64_t>(t);
}

// The same count when T is a normal finite number and the estimate lies in [2, 2^32 - 3] (wave-uniform
// test; otherwise k2_threshold_exact): the estimate is then within 2^-20 of the boundary (two roundings
// in p0 / T 2^32, one in the test's product), so none of the exact routine's clamps, range tests or
// outer tests can fire, and because the test is monotone in u two evaluations decide the count.
__device__ __forceinline__ uint64_t k2_threshold(double p0, double T) {
  const double est = p0 / T * 4294967296.0;
  const bool safe = T >= 1e-280 && T <= 1e280 && est >= 2.0 && est <= 4294967293.0;
  if (!__all(safe)) return k2_threshold_exact(p0, T);
  const double t0 = __builtin_floor(est);
  auto pred = [&](double u) { return u * (1.0 / 4294967296.0) * T < p0; };
  const bool q0 = pred(t0);
  const bool q1 = pred(q0 ? t0 + 1.0 : t0 - 1.0);
  const double t = q0 ? (q1 ? t0 + 2.0 : t0 + 1.0) : (q1 ? t0 : t0 - 1.0);
  return static_cast<uint64_t>(t);
}

// value held by lane J of the caller's quad (v_mov_b32 with DPP quad_perm, full rate)
template <int J> __device__ __forceinline__ double quad_bcast(double v) {
  constexpr int ctrl = J | (J << 2) | (J << 4) | (J << 6);
  const uint64_t u = miso_d2u(v);
  const uint32_t lo = __builtin_amdgcn_mov_dpp(static_cast<int>(u), ctrl, 0xF, 0xF, true);
  const uint32_t hi = __builtin_amdgcn_mov_dpp(static_cast<int>(u >> 32), ctrl, 0xF, 0xF, true);
  return miso_u2d((static_cast<uint64_t>(hi) << 32) | lo);
}

// value held by an arbitrary lane (ds_bpermute): used when a chain's lanes are not quad aligned
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const uint64_t u = miso_d2u(v);
  const uint32_t lo = __shfl(static_cast<int>(u), src_lane);
  const uint32_t hi = __shfl(static_cast<int>(u >> 32), src_lane);
  return miso_u2d((static_cast<uint64_t>(hi) << 32) | lo);
}

// Evaluate routine f on 3 or 4 arguments with NR lanes of the chain working in parallel: the lane
// with role r evaluates argument r, results are broadcast back to every lane of the chain
// (NR == 1: plain scalar calls).  QUAD: the chain's lanes start on a quad boundary (DPP broadcast).
// Written with scalars only -- argument arrays end up on the stack (scratch) in some instantiations.
template <int NR, bool QUAD, int R> __device__ __forceinline__ double role_bcast(double y, int base) {
  if (QUAD && NR == 4) return quad_bcast<R>(y);
  return lane_bcast(y, base + R);
}

template <int NR, bool QUAD, class F>
__device__ __forceinline__ void vec_eval3(F f, double a0, double a1, double a2, double &o0, double &o1,
                                          double &o2, int role, int base) {
  if (NR == 1) { o0 = f(a0); o1 = f(a1); o2 = f(a2); return; }
  if (NR == 2) {
    const double y = f(role == 1 ? a1 : a0);
    o0 = role_bcast<NR, QUAD, 0>(y, base); o1 = role_bcast<NR, QUAD, 1>(y, base);
    o2 = f(a2);
    return;
  }
  const double y = f(role == 1 ? a1 : (role == 2 ? a2 : a0));
  o0 = role_bcast<NR, QUAD, 0>(y, base); o1 = role_bcast<NR, QUAD, 1>(y, base);
  o2 = role_bcast<NR, QUAD, 2>(y, base);
}

template <int NR, bool QUAD, class F>
__device__ __forceinline__ void vec_eval4(F f, double a0, double a1, double a2, double a3, double &o0,
                                          double &o1, double &o2, double &o3, int role, int base) {
  if (NR == 1) { o0 = f(a0); o1 = f(a1); o2 = f(a2); o3 = f(a3); return; }
  if (NR == 2) {
    const double y = f(role == 1 ? a1 : a0), z = f(role == 1 ? a3 : a2);
    o0 = role_bcast<NR, QUAD, 0>(y, base); o1 = role_bcast<NR, QUAD, 1>(y, base);
    o2 = role_bcast<NR, QUAD, 0>(z, base); o3 = role_bcast<NR, QUAD, 1>(z, base);
    return;
  }
  if (NR == 3) {
    const double y = f(role == 1 ? a1 : (role == 2 ? a2 : a0));
    o0 = role_bcast<NR, QUAD, 0>(y, base); o1 = role_bcast<NR, QUAD, 1>(y, base);
    o2 = role_bcast<NR, QUAD, 2>(y, base);
    o3 = f(a3);
    return;
  }
  const double y = f(role == 1 ? a1 : (role == 2 ? a2 : (role == 3 ? a3 : a0)));
  o0 = role_bcast<NR, QUAD, 0>(y, base); o1 = role_bcast<NR, QUAD, 1>(y, base);
  o2 = role_bcast<NR, QUAD, 2>(y, base); o3 = role_bcast<NR, QUAD, 3>(y, base);
}

// Binomial(n, p) of miso_binomial.h by the G lanes of one chain (lanes base .. base + G - 1, this one is `sub`): the
// same draw, bit for bit, as the sequential routine.  Its rejection sampler (BTRS) tries trial t with words 2t, 2t + 1 of
// the chain's word stream and returns the first trial that is accepted: here lane `sub` evaluates trial j G + sub of
// round j and the lowest accepting lane's candidate is the result -- with 4 lanes nearly always one round instead of
// the ~2.5 a wavefront of independent chains needs until its last lane is through.  Small n min(p, q) (inversion):
// every lane for itself.  All lanes of a chain call this together.
template <int G>
__device__ __forceinline__ int32_t binomial_coop(uint64_t seed, uint32_t event_id, uint32_t chain, uint32_t iter,
                                                 int32_t n, double p, const double *__restrict__ lf, int sub, int base) {
  if (G == 1) {
    miso_ustream us;
    miso_ustream_init(&us, seed, event_id, chain, iter, MISO_SITE_COUNTS);
    return miso_binomial(&us, n, p, lf);
  }
  if (n <= 0 || !(p > 0.0)) return 0;
  if (p >= 1.0) return n;
  const double r = p > 0.5 ? 1.0 - p : p;
  int32_t y;
  if (static_cast<double>(n) * r < 10.0) {
    miso_ustream us;
    miso_ustream_init(&us, seed, event_id, chain, iter, MISO_SITE_COUNTS);
    y = miso_binomial_inversion(&us, n, r);
  } else {   // miso_binomial_btrs, G trials per round
    const double q = 1.0 - r, dn = static_cast<double>(n);
    const double spq = miso_det_sqrt(dn * r * q);
    const double b = 1.15 + 2.53 * spq;
    const double aa = -0.0873 + 0.0248 * b + 0.01 * r;
    const double c = dn * r + 0.5;
    const double vr = 0.92 - 4.2 / b;
    const double alpha = (2.83 + 5.1 / b) * spq;
    const double m = __builtin_floor((dn + 1.0) * r);
    const double lpq = miso_det_log(r / q);
    const double h = MISO_LF_AT(lf, static_cast<int32_t>(m)) + MISO_LF_AT(lf, n - static_cast<int32_t>(m));
    const unsigned long long group = (G >= 64 ? ~0ull : ((1ull << G) - 1ull)) << base;
    y = static_cast<int32_t>(m);
    for (int round = 0; round < 4096 / G; round++) {
      const uint32_t t = static_cast<uint32_t>(round * G + sub);
      const miso_u32x4 blk = miso_draw_block(seed, event_id, chain, iter, MISO_SITE_COUNTS, t >> 1);
      const double u = miso_u01((t & 1u) ? blk.v[2] : blk.v[0]) - 0.5;
      double v = miso_u01((t & 1u) ? blk.v[3] : blk.v[1]);
      const double us = 0.5 - __builtin_fabs(u);
      const double k = __builtin_floor((2.0 * aa / us + b) * u + c);
      bool ok = false;
      if (k >= 0.0 && k <= dn) {
        if ((us >= 0.07 && v <= vr) || v == 0.0) ok = true;
        else {
          v = v * alpha / (aa / (us * us) + b);
          ok = miso_det_log(v) <= (h - MISO_LF_AT(lf, static_cast<int32_t>(k)) - MISO_LF_AT(lf, n - static_cast<int32_t>(k))) + (k - m) * lpq;
        }
      }
      const unsigned long long hit = __ballot(ok) & group;
      if (hit) {
        y = __shfl(static_cast<int32_t>(k), __builtin_ctzll(hit));
        break;
      }
    }
  }
  if (y < 0) y = 0;
  if (y > n) y = n;
  return p > 0.5 ? n - y : y;
}

}  // namespace

// WPB = wavefronts per workgroup.  WPB = 8 (single-end): one workgroup fills a CU's eight resident
// slots, wavefronts w and w + 4 share a SIMD, and with a.pair_waves the pair takes the p-th heaviest
// and the p-th lightest group of chains (the slot list is sorted by drawing reads), so every SIMD
// carries the same total work whatever the spread of the events' sizes.
// MODE: 0 single-end; 1 paired-end, any event; 2 paired-end events none of whose drawing reads touches a
// non-finite score (most): the read loop without the "bad score" bookkeeping, fragment indices prefetched.
// (Tried and dropped: per-iteration weight tables w_k[f] = psi_k fp[f] per chain plus a per-read score
// difference -- 17 % fewer VALU per read, but 3.9 KB of LDS per chain instead of 1.9 KB and 8 B instead of
// 4 B per read and iteration from L2 / MALL made it 7 % SLOWER: profiles/r02_pe_k2_modes.txt.)
#ifndef MISO_K2_PE_UNROLL
#define MISO_K2_PE_UNROLL 1
#endif
#ifndef MISO_K2_PE_IDENT
#define MISO_K2_PE_IDENT 1
#endif
// One-round single-end launches (sampler_k2_multi<0, 8>): wavefronts w and w + 4 of the workgroup share a SIMD and the
// launch is over when the slower of the two is.  The SIMD's arbiter serves the OLDER wavefront whenever it can issue
// (profiles/r03_wave_time.txt: it is done at 0.6 of the launch, its partner then runs on alone at a lone wavefront's
// issue rate), so the pair is told to keep step: every iteration a wavefront posts its iteration number, reads its
// partner's and raises its own priority when it is behind (s_setprio: the arbiter looks at the user priority before the
// age).  Chains of one batch run the same number of iterations, so "same iteration" is "same share of the work done".
// Results do not depend on any of this: where wavefronts sit is the hardware's choice (wavefronts w and w + 4 of an eight-wavefront
// workgroup on one SIMD is what this device does, tools/archive/placement_check.py); on another placement the priorities are a
// no-op between wavefronts that do not compete, and the LDS word of a wavefront that left early only makes its partner "ahead".
__shared__ int k2_prog[8];
__device__ __forceinline__ void k2_balance(int wv, int m) {
  __hip_atomic_store(&k2_prog[wv], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const int d = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&k2_prog[wv ^ 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) - m;
  if (d > 0) __builtin_amdgcn_s_setprio(2);
  else if (d < 0) __builtin_amdgcn_s_setprio(0);
  else __builtin_amdgcn_s_setprio(1);
}

typedef const __attribute__((address_space(3))) double *k2_lds_cdp;
typedef const __attribute__((address_space(3))) int32_t *k2_lds_cip;
__device__ __forceinline__ double k2_lds_f64(uint32_t addr) { return *reinterpret_cast<k2_lds_cdp>(static_cast<uintptr_t>(addr)); }
__device__ __forceinline__ int32_t k2_lds_i32(uint32_t addr) { return *reinterpret_cast<k2_lds_cip>(static_cast<uintptr_t>(addr)); }

// The kernel's body: workgroup block_x of grid_x (what blockIdx.x / gridDim.x are for sampler_k2 itself;
// sampler_k2_mix runs two bodies of different lanes per chain side by side in one launch).
// WIDE (G = 64): one chain owns the WHOLE workgroup, 64 x WPB lanes -- events with tens of thousands of drawing
// reads, whose read loop on one wavefront would outlast the rest of the batch.  Every wavefront carries the chain's
// state and runs the MH step redundantly (same inputs, same routines, same bits), the lanes of all wavefronts stride
// over the draw quads, and the per-wavefront counts meet in LDS once per Gibbs step (a.red_off: byte offset of the
// scratch in the dynamic LDS; two buffers, so one barrier per step).
// COLLAPSED (single-end; kernels_lane.hip sampler_k2c): the Gibbs step draws the COUNT of the chain's exchangeable reads
// on isoform 0 as one exact binomial (include/miso_binomial.h) instead of sweeping the reads; the run's last step is
// the sweep, so the returned assignment is a per-read draw.  The chain's G lanes share the work: the Metropolis-Hastings
// step as always, the binomial's rejection trials G at a time (binomial_coop).
// wave_override >= 0 (sampler_k2_multi with a.wave_tab): which wavefront of the run this one is, instead of the place
// block_x / pair_waves give it.
template <int G, int MODE, int WPB, bool WIDE = false, bool COLLAPSED = false>
__device__ __forceinline__ void k2_body(const KernelArgs &a, unsigned block_x, unsigned grid_x, long wave_override = -1) {
  static_assert(!COLLAPSED || (MODE == 0 && !WIDE && (G & (G - 1)) == 0), "collapsed: single-end, 1, 2, 4 ... lanes per chain");
  static_assert(!WIDE || G == 64, "a workgroup-wide chain uses whole wavefronts");
  // WIDE: which chain this workgroup works on, alone or as one of several (coop.hpp; a.coop_tab is indexed by the
  // workgroup's number within its run)
  CoopGroup cg{0, 1, nullptr};
  unsigned wide_chain = block_x;
  if (WIDE && a.coop_tab) {
    const int32_t *t = a.coop_tab + 4 * static_cast<size_t>(block_x);
    wide_chain = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(t[0]));
    cg.rank = __builtin_amdgcn_readfirstlane(t[1]); cg.n = __builtin_amdgcn_readfirstlane(t[2]);
    cg.mem = a.coop_mem + static_cast<size_t>(__builtin_amdgcn_readfirstlane(t[3])) * COOP_WORDS;
    if (a.coop_max_polls) cg.max_polls = a.coop_max_polls;
  }
  uint32_t coop_step = 0; bool coop_ok = true;
  const int GE = WIDE ? 64 * WPB * cg.n : G;       // lanes striding over one chain's draw quads
  constexpr bool PE = MODE != 0;
  constexpr bool PEW = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_k2[];
  double *lds_fp = reinterpret_cast<double *>(smem_k2);  // PE: fragment-length probabilities
  const int tab_n = PEW ? pe_k2_entries(a.il) : 2 * a.il;   // entries of the score table (and, MODE 2, of the probabilities)
  if (PE) {
    for (int i = threadIdx.x; i < a.il; i += blockDim.x) {
      const double v = a.frag_prob[i];
      lds_fp[i] = v;
      if (PEW) lds_fp[a.il + i] = v;               // MODE 2 (device.hpp pe_k2_entries): [fp, fp, 0.0, 1.0]
    }
    if (PEW && threadIdx.x == 0) { lds_fp[2 * a.il] = 0.0; lds_fp[2 * a.il + 1] = 1.0; }
    __syncthreads();                               // the only block-level barrier
  }
  constexpr int CPW = 64 / G;                      // chains per wavefront
  constexpr int NR = G >= 4 ? 4 : G;               // lanes cooperating on the scalar math
  constexpr bool QUAD = (G % 4) == 0;
  constexpr bool POW2 = (G & (G - 1)) == 0;
#ifdef MISO_K2_WAVETIME   // tools/wave_time.py: how long every wavefront ran, through ChainStats::hw_id (diagnostic build)
  uint64_t wt_t0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt_t0) : : "memory");   // 100 MHz; "memory": stays where it is
#endif
  const int lane = threadIdx.x & 63;
  const int grp_raw = lane / G;
  const bool lane_used = grp_raw < CPW;            // 64 % G lanes at the top of the wave idle
  const int grp = lane_used ? grp_raw : CPW - 1;
  const int base_lane = grp * G;
  const int sub = WIDE ? cg.rank * 64 * WPB + static_cast<int>(threadIdx.x) : (lane_used ? lane - base_lane : 0);
  const int lsub = WIDE ? lane : sub;              // position among the chain's lanes of THIS wavefront
  const int role = lsub % NR;
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  long wave_id = WIDE ? static_cast<long>(wide_chain) : static_cast<long>(block_x) * WPB + (threadIdx.x >> 6);
  if (!WIDE && WPB == 8 && a.pair_waves) {
    const int w = threadIdx.x >> 6;
    const long p = 4 * static_cast<long>(block_x) + (w & 3);             // pair index: heaviest first
    wave_id = (w < 4) ? p : static_cast<long>(grid_x) * 8 - 1 - p;       // ... with the p-th lightest
  }
  if (!WIDE && wave_override >= 0) wave_id = wave_override;
  if (wave_id * CPW >= n_chains) return;  // whole wavefront idle
  long slot = wave_id * CPW + grp;
  const bool live = lane_used && slot < n_chains;  // dead lanes shadow a chain and store nothing
  if (slot >= n_chains) slot = n_chains - 1;

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
  // full blocks of draws and the draws of the partial one: a Philox block is four reads paired-end, EIGHT single-end (one
  // half-word each: the lazy low bits of include/miso_philox.h)
  const int nfq = PE ? n_draw >> 2 : n_draw >> 3, rem = PE ? n_draw & 3 : n_draw & 7;
  // trips of the Gibbs loop (UQ quads per lane per trip); must be wave-uniform
  constexpr int UQ = PE ? 2 : MISO_K2_UQ;
  int trips = (nfq + UQ * GE - 1) / (UQ * GE);
  int any_rem = rem;
  for (int off = 32; off >= 1; off >>= 1) {
    trips = max(trips, __shfl_xor(trips, off));
    any_rem |= __shfl_xor(any_rem, off);
  }
  trips = __builtin_amdgcn_readfirstlane(trips);     // wave-uniform by construction: say so
  any_rem = __builtin_amdgcn_readfirstlane(any_rem);

  double *samples = reinterpret_cast<double *>(a.out_pool + E.off_samples);
  double *loglik = reinterpret_cast<double *>(a.out_pool + E.off_loglik);
  uint8_t *drawass = a.out_pool + E.off_drawass;
  int32_t *trace = (E.off_trace == NO_TRACE) ? nullptr
                                             : reinterpret_cast<int32_t *>(a.out_pool + E.off_trace);
  const uint32_t k0 = static_cast<uint32_t>(a.seed), k1 = static_cast<uint32_t>(a.seed >> 32);
  const uint32_t c2_gibbs = MISO_SITE_GIBBS | (chain << 8), c2_mh = MISO_SITE_MH | (chain << 8);
  const GibbsRng rng = gibbs_rng_init(a.seed, event_id, chain);

  int cnt0 = 0, cnt1 = 0;
  int64_t rfix = 0; int rbad = 0;   // PE: fixed-point sum of the assigned reads' fragment scores
  const uint4 *fragq = reinterpret_cast<const uint4 *>(a.in_pool + E.off_draw);  // PE: 4 reads x (f0 | f1 << 16)
  // PE: the event's fixed-point score table (2 x il int32) sits in the chain's LDS slice: a per-read
  // gather from L2 would cost more than the whole rest of the Gibbs step
  int32_t *lds_tab = reinterpret_cast<int32_t *>(smem_k2 + (((PEW ? tab_n : a.il) * 8 + 15) & ~15)) +
                     (static_cast<size_t>(threadIdx.x >> 6) * CPW + grp) * tab_n;
  if (PE) {
    const int32_t *sfix = reinterpret_cast<const int32_t *>(a.in_pool + E.off_sfix);
    for (int i = lsub; i < tab_n; i += G) lds_tab[i] = i < 2 * a.il ? sfix[i] : 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  // MODE 2 reads the LDS by byte address (no symbol arithmetic, never a flat access): the dynamic LDS -- the
  // only LDS of this kernel -- starts at 0
  if (PEW && static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem_k2)) != 0u) __builtin_trap();
  const uint32_t tab_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_tab));
  const uint4 *denseq = reinterpret_cast<const uint4 *>(a.in_pool + (PEW ? E.off_dense : E.off_draw));
  PsiTerms cur;
  double alpha = 0.0;

  // paired-end pick of one read (miso_paired.c:11-22, 64-68): weights psi_k * fragProb(frag_k)
  auto pe_pick = [&](uint32_t ff, uint32_t uword, int64_t &acc, int &bad) __attribute__((always_inline)) {
    const uint32_t f0 = ff & 0xFFFFu, f1 = ff >> 16;
    const double c0 = 0.0 + cur.x0 * lds_fp[f0];
    const double T = c0 + cur.x1 * lds_fp[f1];
    const bool p0 = miso_u01(uword) * T < c0;
    const int32_t v = lds_tab[p0 ? f0 : a.il + f1];
    const bool isbad = v == SFIX_BAD;
    bad |= isbad ? 1 : 0;
    acc += isbad ? 0 : v;
    return p0;
  };

  // WIDE: the wavefronts' partial sums meet in LDS (buffer `red_par`, flipped every Gibbs step)
  int red_par = 0;
  auto wide_sum = [&](int &d0, int64_t &acc, int &bad) __attribute__((always_inline)) {
    struct Red { int64_t acc; int d0, bad; };
    Red *red = reinterpret_cast<Red *>(smem_k2 + a.red_off) + red_par * WPB;
    if (lane == 0) red[threadIdx.x >> 6] = Red{acc, d0, bad};
    __syncthreads();
    int td = 0, tb = 0; int64_t ta = 0;
#pragma unroll
    for (int w = 0; w < WPB; w++) { const Red r = red[w]; td += r.d0; ta += r.acc; tb |= r.bad; }
    red_par ^= 1;
    if (cg.n > 1) {   // ... and the chain's other workgroups' (coop.hpp): accumulator coop_step % 3, one barrier
      uint32_t *accp = coop_acc(cg, coop_step);
      if (threadIdx.x < 64) {
        if (lane == 0) {
          if (td) atomicAdd(&accp[4], static_cast<uint32_t>(td));
          if (PE) { atomicAdd(reinterpret_cast<unsigned long long *>(accp), static_cast<unsigned long long>(ta)); if (tb) atomicOr(&accp[2], 1u); }
        }
        if (cg.rank == 0 && lane < COOP_ACC) atomicExch(&coop_acc(cg, coop_step + 1)[lane], 0u);   // next step's, read last two steps ago
        __threadfence();   // this wavefront's atomics are at the L2 before the workgroup announces its arrival
      }
      int *flag = reinterpret_cast<int *>(smem_k2 + a.red_off + 2 * WPB * 16);
      coop_ok = coop_barrier(cg, coop_step, flag) && coop_ok;
      td = static_cast<int>(coop_load(&accp[4]));
      if (PE) {
        ta = static_cast<int64_t>((static_cast<uint64_t>(coop_load(&accp[1])) << 32) | coop_load(&accp[0]));
        tb = static_cast<int>(coop_load(&accp[2]));
      }
      coop_step++;
    }
    d0 = td; acc = ta; bad = tb;
  };

  // Gibbs step for the current psi (miso.c:30-91 restricted to two compatible isoforms)
#ifdef MISO_K2_PROFILE
  uint64_t pf_mh = 0, pf_thr = 0, pf_loop = 0, pf_red = 0, pf_rec = 0;
#endif
  auto gibbs = [&](uint32_t iter) {
    if (PEW) {
      // MODE 2: no drawing read of the event touches a non-finite score (host.cpp pe_delta): no "bad"
      // bookkeeping; dense records (device.hpp pe_k2_entries): one index per (read, isoform) serves the
      // probability and the score, lanes beyond the last quad work on the quad of padding reads behind it
      // (weight 0 against a positive one: never isoform 0, score 0) -- no range tests; the next quad's
      // record is fetched one trip ahead.  Per read: two SDWA shifts, two LDS reads by byte address, the
      // reference's arithmetic (miso_paired.c:11-22, 64-68), one select, one add-with-carry for the count,
      // one 64-bit multiply-add for the score sum.
      const uint32_t n0r0 = rng.p1hi ^ iter ^ k0;
      const int nq = (n_draw + 3) >> 2;
      const double x0 = cur.x0, x1 = cur.x1;
      int d0 = 0; int64_t acc = 0;
      const uint32_t three = 3u;
      auto rec_at = [&](int q) __attribute__((always_inline)) { return denseq[(lane_used && q < nq) ? q : nq]; };
      uint4 fn = rec_at(sub);
#pragma unroll MISO_K2_PE_UNROLL
      for (int j = 0; j < 2 * trips + 1; j++) {
        const int q = sub + j * GE;
        const uint4 f = fn;
        fn = rec_at(q + GE);
        const miso_u32x4 u = philox_gibbs<true>(rng, static_cast<uint32_t>(q), n0r0);
        const uint32_t ff[4] = {f.x, f.y, f.z, f.w};
        uint32_t o0[4], sel[4]; double c0[4], rnd[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          uint32_t o1;   // 8 x index
          asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(o0[r]) : "v"(three), "v"(ff[r]));
          asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(o1) : "v"(three), "v"(ff[r]));
          // (Dropping the "0.0 +" -- an identity here -- was measured 6 % SLOWER in the loop's first form;
          // the add seems to give the scheduler a better order.)
#if MISO_K2_PE_IDENT
          c0[r] = 0.0 + x0 * k2_lds_f64(o0[r]);
#else
          c0[r] = x0 * k2_lds_f64(o0[r]);
#endif
          const double T = c0[r] + x1 * k2_lds_f64(o1);
          rnd[r] = miso_u01(u.v[r]) * T;
          sel[r] = o1;
        }
        {   // the four compares first: a VALU read of an SGPR must stay two instructions behind the compare that wrote it
          uint64_t m0, m1, m2, m3, junk;
          asm("v_cmp_lt_f64_e64 %5, %10, %14\n\tv_cmp_lt_f64_e64 %6, %11, %15\n\tv_cmp_lt_f64_e64 %7, %12, %16\n\tv_cmp_lt_f64_e64 %8, %13, %17\n\t"
              "v_cndmask_b32_e64 %0, %0, %18, %5\n\tv_addc_co_u32_e64 %4, %9, %4, 0, %5\n\t"
              "v_cndmask_b32_e64 %1, %1, %19, %6\n\tv_addc_co_u32_e64 %4, %9, %4, 0, %6\n\t"
              "v_cndmask_b32_e64 %2, %2, %20, %7\n\tv_addc_co_u32_e64 %4, %9, %4, 0, %7\n\t"
              "v_cndmask_b32_e64 %3, %3, %21, %8\n\tv_addc_co_u32_e64 %4, %9, %4, 0, %8"
              : "+v"(sel[0]), "+v"(sel[1]), "+v"(sel[2]), "+v"(sel[3]), "+v"(d0), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3), "=&s"(junk)
              : "v"(rnd[0]), "v"(rnd[1]), "v"(rnd[2]), "v"(rnd[3]), "v"(c0[0]), "v"(c0[1]), "v"(c0[2]), "v"(c0[3]),
                "v"(o0[0]), "v"(o0[1]), "v"(o0[2]), "v"(o0[3]));
        }
        int32_t v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = k2_lds_i32(tab_addr + (sel[r] >> 1));
#pragma unroll
        for (int r = 0; r < 4; r++) {
          uint64_t junk;
          asm("v_mad_i64_i32 %0, %1, %2, 1, %0" : "+v"(acc), "=s"(junk) : "v"(v[r]));
        }
      }
      if (POW2) {
#pragma unroll
        for (int off = G >> 1; off >= 1; off >>= 1) { d0 += __shfl_xor(d0, off); acc += __shfl_xor(acc, off); }
      } else {
        int tot = 0; int64_t ta = 0;
#pragma unroll
        for (int j = 0; j < G; j++) { tot += __shfl(d0, base_lane + j); ta += __shfl(acc, base_lane + j); }
        d0 = tot; acc = ta;
      }
      if (WIDE) { int nobad = 0; wide_sum(d0, acc, nobad); }
      cnt0 = base0 + d0;
      cnt1 = base1 + (n_draw - d0);
      rfix = E.base_sfix + acc;
      rbad = E.base_bad;
      return;
    }
    if (PE) {
      const uint32_t n0r0 = rng.p1hi ^ iter ^ k0;
      const int nq = (n_draw + 3) >> 2;
      int d0 = 0, bad = 0; int64_t acc = 0;
      // No data-dependent branches: a quad beyond the chain's reads is loaded from a clamped address
      // and its reads are masked, so the four reads' LDS gathers are all in flight at once (the
      // guarded form waited on each read's tables separately: 16 s_waitcnt per quad) and the round
      // keys are rebuilt per block instead of spilled (gibbs_rng.hpp).
      const int q_last = max(nq - 1, 0);
      const double x0 = cur.x0, x1 = cur.x1;
      for (int j = 0; j < 2 * trips + 1; j++) {
        const int q = sub + j * GE;
        const miso_u32x4 u = philox_gibbs<true>(rng, static_cast<uint32_t>(q), n0r0);
        const uint4 f = fragq[min(q, q_last)];
        const int left = (q < nq && lane_used) ? n_draw - 4 * q : 0;   // reads of this quad that exist
        const uint32_t ff[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const bool valid = left > r;
          const uint32_t fv = valid ? ff[r] : 0u;
          const uint32_t f0 = fv & 0xFFFFu, f1 = fv >> 16;
          const double c0 = 0.0 + x0 * lds_fp[f0];               // miso_paired.c:11-22, 64-68
          const double T = c0 + x1 * lds_fp[f1];
          const bool p0 = miso_u01(u.v[r]) * T < c0;
          const int32_t v = lds_tab[p0 ? f0 : a.il + f1];
          const bool isbad = v == SFIX_BAD;
          bad |= (valid & isbad) ? 1 : 0;
          acc += (valid & !isbad) ? v : 0;
          d0 += (valid & p0) ? 1 : 0;
        }
      }
      if (POW2) {
#pragma unroll
        for (int off = G >> 1; off >= 1; off >>= 1) {
          d0 += __shfl_xor(d0, off); acc += __shfl_xor(acc, off); bad |= __shfl_xor(bad, off);
        }
      } else {
        int tot = 0, tb = 0; int64_t ta = 0;
#pragma unroll
        for (int j = 0; j < G; j++) {
          tot += __shfl(d0, base_lane + j); ta += __shfl(acc, base_lane + j); tb |= __shfl(bad, base_lane + j);
        }
        d0 = tot; acc = ta; bad = tb;
      }
      if (WIDE) wide_sum(d0, acc, bad);
      cnt0 = base0 + d0;
      cnt1 = base1 + (n_draw - d0);
      rfix = E.base_sfix + acc;
      rbad = bad | E.base_bad;
      return;
    }
    if constexpr (COLLAPSED) {
      if (iter != (a.M > 0 ? static_cast<uint32_t>(a.M - 1) : MISO_ITER_INIT)) {   // every step but the run's last
        const int d0 = binomial_coop<G>(a.seed, event_id, chain, iter, n_draw, cur.x0 / ((0.0 + cur.x1) + cur.x0),
                                        a.logfact, sub, base_lane);
        cnt0 = base0 + d0;
        cnt1 = base1 + (n_draw - d0);
        return;
      }
    }
    PROF_T(g0);
    // u < t with u = hi 2^16 + lo:  hi < t >> 16, or hi == t >> 16 and lo < (t & 0xFFFF).  The loop looks at the high
    // halves only -- eight reads per Philox block -- and notes the trip in which one of them EQUALS t >> 16 (one read
    // in 65 536); that trip's blocks' low halves (site MISO_SITE_GIBBS_LOW) are drawn behind the loop, by the lane that owns them.
    const uint64_t t = k2_threshold(cur.x0, (0.0 + cur.x0) + cur.x1);
    const uint32_t th = static_cast<uint32_t>(t >> 16), tl = static_cast<uint32_t>(t) & 0xFFFFu;   // th <= 65536
    const uint32_t n0r0 = rng.p1hi ^ iter ^ k0;
    int d0 = 0, amb_n = 0; uint32_t amb_t = 0; bool amb_p = false;   // trips of this lane with a high half on the threshold, the last of them; the partial block
    PROF_T(g1);
    PROF_ADD(pf_thr, g0, g1);
    auto halves = [&](const miso_u32x4 &u, int nh, int &below, int &equal) __attribute__((always_inline)) {
      below = 0; equal = 0;
#pragma unroll
      for (int h = 0; h < 8; h++) {
        const uint32_t x = (h & 1) ? (u.v[h >> 1] >> 16) : (u.v[h >> 1] & 0xFFFFu);
        below += (h < nh && x < th) ? 1 : 0;
        equal |= (h < nh && x == th) ? 1 : 0;
      }
    };
    // Both half-words of a generator word at once (packed 16-bit arithmetic, no per-half compare into a lane mask):
    //   below: saturating th - x is non-zero iff x < th; min(.., 1) is the count;   equal: x ^ th is zero iff x == th,
    // kept as the running minimum over the trip's words and tested once per trip (a 32-bit word has a zero half iff
    // (m - 0x00010001) & ~m & 0x80008000).  th = 65536 (t = 2^32: every read picks isoform 0) does not fit a half:
    // counted in closed form behind the loop.
    // (inline assembly: written with vector types the compiler recognises the idiom and goes back to one compare per half.
    // Measured, same box, 40 000 events x 1000 reads: one SDWA compare per half into a lane mask + add-with-carry, 110
    // VALU per 16 reads, 74.6 ms; this form, 108 VALU but no lane masks, 71.0 ms; MISO defaults 235.9 -> 219.1 ms,
    // hg19-like read counts 66.4 -> 59.4 ms; the equal test once per trip instead of once per block: 70.9 -> 69.3 ms,
    // defaults 219.5 -> 207.9 ms; profiles/r04_lazy_low_bits.txt)
    const uint32_t thc = th > 0xFFFFu ? 0xFFFFu : th;
    const uint32_t T2 = thc | (thc << 16), one2 = 0x00010001u;
    uint32_t accv = 0;
    // (the two 16-bit counters of a lane take 4 per block each: emptied every 16000 blocks, which only a chain of more
    // than 10^5 reads forced onto a single lane ever reaches)
    constexpr int CHUNK = 16000 / UQ;
    for (int j0 = 0; j0 < trips; j0 += CHUNK) {
      const int j1 = min(trips, j0 + CHUNK);
      for (int j = j0; j < j1; j++) {
        miso_u32x4 u[UQ];
#pragma unroll
        for (int i = 0; i < UQ; i++)
          u[i] = philox_gibbs(rng, static_cast<uint32_t>(sub + (UQ * j + i) * GE), n0r0);
        uint32_t m = u[0].v[0] ^ T2;   // the trip's running minimum of x ^ th, both halves (blocks beyond the chain's
                                       // last included: a flag too many only costs the look behind the loop)
#pragma unroll
        for (int i = 0; i < UQ; i++) {
          const int q = sub + (UQ * j + i) * GE;
          uint32_t c = 0;
#pragma unroll
          for (int w = 0; w < 4; w++) {
            uint32_t d;
            asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(d) : "v"(T2), "v"(u[i].v[w]));
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(d), "v"(one2));
            asm("v_pk_add_u16 %0, %1, %2" : "=v"(c) : "v"(c), "v"(d));
            if (i == 0 && w == 0) continue;
            const uint32_t y = u[i].v[w] ^ T2;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(m), "v"(y));
          }
          uint32_t cm = (q < nfq) ? c : 0u;
          asm("v_pk_add_u16 %0, %1, %2" : "=v"(accv) : "v"(accv), "v"(cm));
        }
        if (((m - 0x00010001u) & ~m & 0x80008000u) != 0u) { amb_t = static_cast<uint32_t>(j); amb_n++; }
      }
      d0 += static_cast<int>(accv & 0xFFFFu) + static_cast<int>(accv >> 16);
      accv = 0;
    }
    if (th > 0xFFFFu) d0 = (sub < nfq) ? 8 * ((nfq - 1 - sub) / GE + 1) : 0;
    if (any_rem) {  // the partial block, owned by one lane of the group
      const miso_u32x4 u = philox_gibbs(rng, static_cast<uint32_t>(nfq), n0r0);
      int cp, eq;
      halves(u, rem, cp, eq);
      const bool mine = sub == (nfq % GE);
      d0 += mine ? cp : 0;
      amb_p = mine && eq;
    }
    if (!lane_used) { d0 = 0; amb_n = 0; amb_p = false; }
    const bool settle_all = a.pe_force_exact != 0;   // tests: every lane takes the rescan below at every step
    if (tl != 0 && __builtin_expect(settle_all || __any(amb_n != 0 || amb_p), 0)) {
      // the reads whose high half sits ON the threshold: their low halves decide (one wavefront step in four at 1000
      // reads and 16 chains per wavefront; two Philox blocks then)
      auto settle = [&](uint32_t q) {
        const miso_u32x4 hi = miso_philox4x32(q, iter, c2_gibbs, event_id, k0, k1);
        const miso_u32x4 lo = miso_philox4x32(q, iter, MISO_SITE_GIBBS_LOW | (chain << 8), event_id, k0, k1);
        const int nh = (static_cast<int>(q) == nfq) ? rem : 8;
        int more = 0;
        for (int h = 0; h < nh; h++)
          more += (miso_block_half(hi, h) == th && miso_block_half(lo, h) < tl) ? 1 : 0;
        return more;
      };
      // a full block of this lane: its low halves if one of its high halves is on the threshold
      auto look = [&](int q) {
        if (q >= nfq) return 0;
        const miso_u32x4 hi = miso_philox4x32(static_cast<uint32_t>(q), iter, c2_gibbs, event_id, k0, k1);
        bool on = false;
        for (int h = 0; h < 8; h++) on |= miso_block_half(hi, h) == th;
        return on ? settle(static_cast<uint32_t>(q)) : 0;
      };
      // the partial block (its owner only; settle() counts nothing when no high half is on the threshold)
      if (lane_used && rem != 0 && sub == (nfq % GE) && (amb_p || settle_all)) d0 += settle(static_cast<uint32_t>(nfq));
      if (amb_n == 1 && !settle_all) {        // the full blocks: one trip's ...
        for (int i = 0; i < UQ; i++) d0 += look(sub + (UQ * static_cast<int>(amb_t) + i) * GE);
      } else if ((amb_n > 1 || settle_all) && lane_used) {   // ... or, with several such trips, all of the lane's
        for (int q = sub; q < nfq; q += GE) d0 += look(q);
      }
    }
    PROF_T(g2);
    PROF_ADD(pf_loop, g1, g2);
    if (POW2) {
#pragma unroll
      for (int off = G >> 1; off >= 1; off >>= 1) d0 += __shfl_xor(d0, off);
    } else {
      int tot = 0;
#pragma unroll
      for (int j = 0; j < G; j++) tot += __shfl(d0, base_lane + j);
      d0 = tot;
    }
    if (WIDE) { int64_t noacc = 0; int nobad = 0; wide_sum(d0, noacc, nobad); }
    cnt0 = base0 + d0;
    cnt1 = base1 + (n_draw - d0);
    PROF_T(g3);
    PROF_ADD(pf_red, g2, g3);
  };

  // the per-read picks of one Gibbs step, written once for the caller (miso.c:943-946)
  auto gibbs_write = [&](uint32_t iter) {
    const uint64_t t = PE ? 0 : k2_threshold(cur.x0, (0.0 + cur.x0) + cur.x1);
    if (!PE) {   // the same picks read by read, both halves of every uniform (include/miso_philox.h miso_split_word)
      for (int r = sub; r < n_draw; r += GE)
        drawass[r] = (static_cast<uint64_t>(miso_split_word(a.seed, event_id, chain, iter, static_cast<uint32_t>(r))) < t) ? 0 : 1;
      return;
    }
    const int nq = (n_draw + 3) >> 2;
    for (int q = sub; q < nq; q += GE) {
      const miso_u32x4 u = miso_philox4x32(static_cast<uint32_t>(q), iter, c2_gibbs, event_id, k0, k1);
      const uint4 f = PE ? fragq[q] : make_uint4(0, 0, 0, 0);
      const uint32_t ff[4] = {f.x, f.y, f.z, f.w};
      for (int j = 0; j < 4; j++) {
        if (4 * q + j >= n_draw) continue;
        int64_t dummy = 0; int db = 0;
        const bool p0 = PE ? pe_pick(ff[j], u.v[j], dummy, db) : (static_cast<uint64_t>(u.v[j]) < t);
        drawass[4 * q + j] = p0 ? 0 : 1;
      }
    }
  };

  // the iteration's MH-site draws: the accept word and the proposal's standard normal (random.c:1543-1551).
  // They depend on (seed, event, chain, iteration) only, not on the chain's state, so the NR lanes of a
  // chain compute them for NR consecutive iterations at once (lane role r: iteration m0 + r) -- one
  // Philox block + qnorm per NR iterations per lane instead of one per iteration on every lane.
  auto mh_draws = [&](uint32_t iter, double &z, uint32_t &accept_word) {
    const miso_u32x4 b = miso_philox4x32(0u, iter, c2_mh, event_id, k0, k1);
    accept_word = b.v[0];
    z = miso_det_norm_from_unif(miso_u01(b.v[2]), miso_u01(b.v[3]));
  };
  // alpha' = alpha + sd z, psi' = logit_inv(alpha') (miso.c:449-471)
#if MISO_K2_TAB_REGS
  double TE[12], TL[12];
  det_tables_to_registers(TE, TL);
  auto k2_exp = [&](double v) { return det_exp_t(v, TE); };
  auto k2_log = [&](double v) { return det_log_t(v, TL); };
#else
  auto k2_exp = [](double v) { return miso_det_exp(v); };
  auto k2_log = [](double v) { return miso_det_log(v); };
#endif
  auto propose = [&](double z, double &alphaN, double &x0, double &x1) {
    alphaN = alpha + c.sd * z;
    const double e = k2_exp(alphaN);
    const double sumexp = (0.0 + e) + 1.0;
    x0 = e / sumexp;
    x1 = 1 - (0.0 + x0);
  };

  // ---- initial state (miso.c:362-369 K == 2: alpha = 0; miso.c:834, 841) ----
  {
    double aN, x0, x1, z; uint32_t w;
    mh_draws(MISO_ITER_INIT, z, w);
    propose(z, aN, x0, x1);
    alpha = aN;
    cur = psi_terms(x0, x1, c.cst0, c.cst1);
  }
  gibbs(MISO_ITER_INIT);
  if (a.M == 0 && live && chain == 0) gibbs_write(MISO_ITER_INIT);

  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;
  const bool writer = live && sub == 0;
  // WIDE, eight wavefronts: wavefronts w and w + 4 of the workgroup share a SIMD and belong to the SAME chain.  Both used to run
  // the Metropolis-Hastings step (same inputs, same bits) -- on a SIMD whose two wavefronts are one chain's that is the step
  // twice per iteration, and the launch's largest event (80 000 drawing reads on one workgroup) set the launch's duration: 1.33
  // x every other SIMD's (profiles/r06_wave_time_hg19.txt).  Now the first four wavefronts run the step and publish psi -- all the
  // Gibbs step needs -- through LDS; the other four wait at a barrier (a waiting wavefront issues nothing) and join the read loop.
  const bool helper = WIDE && WPB == 8 && a.wide_dedup && (threadIdx.x >> 6) >= 4;
  double *wide_pub = reinterpret_cast<double *>(smem_k2 + a.red_off + 2 * WPB * 16 + 16);

  double zbuf = 0.0; uint32_t awbuf = 0u;   // this lane's share of NR consecutive iterations' MH draws
  // The proposal of iteration m is made -- alpha' = alpha + sd z_m and exp(alpha') (miso.c:449-471) -- at the END of
  // iteration m - 1, for both outcomes of that iteration's test, in the routine call that evaluates the test's own
  // exponential (below): one call of the exponential instead of two per iteration.  Same inputs, same routine, same bits.
  // Three and more lanes per chain only (SPEC): one or two lanes per chain have no idle lane to give the extra arguments to,
  // and their step stays as it was -- the proposal at the top of its iteration, four exponentials in the softmax pass
  // (measured: the merged form on one lane per chain, MISO's default settings, 199.0 -> 206.6 ms).
  constexpr bool SPEC = NR >= 3;
  double alphaN_p = 0.0, e_p = 0.0;
  if constexpr (SPEC) {
    mh_draws(static_cast<uint32_t>(role), zbuf, awbuf);   // iterations 0 .. NR - 1
    const double z = lane_bcast(zbuf, base_lane);
    alphaN_p = alpha + c.sd * z;
    e_p = k2_exp(alphaN_p);
  }
  const double E0 = k2_exp(0.0);   // exp(maxv - maxv) of the log-sum-exp below
  RoundOpen ro(a);
  for (int m = 0; m < a.M; m++) {
    const bool opens = ro.at(a, m);   // a round's first iteration: no proposal terms (miso.c:866)
    prio_by_progress(a, m);
    if (MODE == 0 && WPB == 8 && !WIDE && !COLLAPSED && a.balance == 1) k2_balance(threadIdx.x >> 6, m);
    if (WIDE && !coop_ok) return;   // the chain's workgroups gave up waiting for each other (coop.hpp): the host reports it
    if (!helper) {
    hash = (hash ^ static_cast<uint32_t>(cnt0)) * 0x100000001B3ull;
    hash = (hash ^ static_cast<uint32_t>(cnt1)) * 0x100000001B3ull;
    if (trace && writer) {
      int32_t *row = trace + (static_cast<size_t>(m) * a.C + chain) * 2;
      row[0] = cnt0; row[1] = cnt1;
    }
    PROF_T(m0);
    double alphaN, x0, x1, zn = 0.0; uint32_t accept_word;
    if constexpr (SPEC) {
      const int ph = m % NR;
      accept_word = static_cast<uint32_t>(__shfl(static_cast<int>(awbuf), base_lane + ph));
      alphaN = alphaN_p;
      const double sumexp = (0.0 + e_p) + 1.0;
      x0 = e_p / sumexp;
      x1 = 1 - (0.0 + x0);
      // z of iteration m + 1 (the draws depend on (seed, event, chain, iteration) only)
      if (ph == NR - 1) mh_draws(static_cast<uint32_t>(m + 1 + role), zbuf, awbuf);   // iterations m + 1 .. m + NR
      zn = lane_bcast(zbuf, base_lane + (ph == NR - 1 ? 0 : ph + 1));
    } else {
      double z;
      if (NR == 1) {
        mh_draws(static_cast<uint32_t>(m), z, accept_word);
      } else {
        const int ph = m % NR;
        if (ph == 0) mh_draws(static_cast<uint32_t>(m + role), zbuf, awbuf);   // iterations m .. m + NR - 1
        z = lane_bcast(zbuf, base_lane + ph);
        accept_word = static_cast<uint32_t>(__shfl(static_cast<int>(awbuf), base_lane + ph));
      }
      propose(z, alphaN, x0, x1);
    }
    PsiTerms nw;
    double ptoCS, ctoPS;
    {
      // NR lanes, one routine, NR arguments per call (see vec_eval)
      auto f_log = k2_log;
      auto f_exp = k2_exp;
      const double ltheta = 1.0 - x0;
      double lgtN;
      vec_eval3<NR, QUAD>(f_log, x0, x1, x0 / ltheta, nw.lx0, nw.lx1, lgtN, role, base_lane);
      nw.x0 = x0; nw.x1 = x1; nw.lgt = lgtN;
      nw.pr = 1.0 / (1.0 * x0) / ltheta;
      const double lp0 = nw.lx0 + c.cst0, lp1 = nw.lx1 + c.cst1;
      const double maxv = (lp1 > lp0) ? lp1 : lp0;  // miso.c:137-140: maxv starts at entry 0
      // of exp(lp0 - maxv), exp(lp1 - maxv) one is exp(x - x) = exp(+0) = E0 for every finite x: three exponentials
      double ex0, ex1, xp, xc;
      {   // (every lanes-per-chain layout: with one lane per chain that is three exponentials instead of four)
        const bool m1 = lp1 > lp0;
        const double dmin = m1 ? lp0 - maxv : lp1 - maxv, dmax = m1 ? lp1 - maxv : lp0 - maxv;
        double emin;
        vec_eval3<NR, QUAD>(f_exp, dmin,
                            prop_exponent(cur.lgt, alphaN, c.sigma),   // theta = psi,  mu = alpha'
                            prop_exponent(nw.lgt, alpha, c.sigma),     // theta = psi', mu = alpha
                            emin, xp, xc, role, base_lane);
        double emax = E0;
        if (__builtin_expect(__any(!(dmax == 0.0)), 0)) emax = f_exp(dmax);   // a non-finite log psi (inf - inf): as written
        ex0 = m1 ? emin : emax; ex1 = m1 ? emax : emin;
      }
      double ls;
      vec_eval3<NR, QUAD>(f_log, (0.0 + ex0) + ex1, c.covar * cur.pr * xp, c.covar * nw.pr * xc, ls, ptoCS,
                          ctoPS, role, base_lane);
      const double lse = ls + maxv;
      nw.lpn0 = lp0 - lse;
      nw.lpn1 = lp1 - lse;
    }
    const double rp = PE ? (rbad ? miso_u2d(0x7FF8000000000000ull)
                                 : static_cast<double>(rfix) * (1.0 / MISO_SFIX_SCALE))
                         : 0.0;
    const double pp = joint<PE>(nw, cnt0, cnt1, c, rp);
    const double pc = joint<PE>(cur, cnt0, cnt1, c, rp);
    // the test's exponential and -- three and more lanes per chain: in the same call, for both outcomes of the test --
    // the next proposal's
    const double targ = !opens ? pp + ptoCS - (pc + ctoPS) : pp - pc;
    double acceptP, aA = 0.0, aR = 0.0, eA = 0.0, eR = 0.0;
    if constexpr (SPEC) {
      aA = alphaN + c.sd * zn; aR = alpha + c.sd * zn;
      vec_eval3<NR, QUAD>(k2_exp, targ, aA, aR, acceptP, eA, eR, role, base_lane);
    } else {
      acceptP = k2_exp(targ);
    }
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    double cJS = pc;
    if (acc) { cur = nw; alpha = alphaN; cJS = pp; accepted++; }
    if constexpr (SPEC) { alphaN_p = acc ? aA : aR; e_p = acc ? eA : eR; }
    PROF_T(m1);
    PROF_ADD(pf_mh, m0, m1);

    // miso.c:882-893.  (Round 5 tried issuing these stores BEHIND the Gibbs step -- vmcnt counts loads and stores in order,
    // so a load issued after a store is only "there" once the store has been acknowledged: no gain paired-end (198.4 vs
    // 199.4 ms), 4 % slower at MISO's default settings, profiles/r05_store_after.txt.)
    if (m >= a.B) {
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
    PROF_T(m2);
    PROF_ADD(pf_rec, m1, m2);
    }   // !helper
    if (WIDE && WPB == 8 && a.wide_dedup) {   // (the next write is behind the Gibbs step's own barrier: one buffer is enough)
      if (threadIdx.x == 0) { wide_pub[0] = cur.x0; wide_pub[1] = cur.x1; }
      __syncthreads();
      if (helper) { cur.x0 = wide_pub[0]; cur.x1 = wide_pub[1]; }
    }
    gibbs(static_cast<uint32_t>(m));
  }
#ifdef MISO_K2_PROFILE
  if (writer && chain == 0 && a.M > 8) {  // smuggle the phase cycle counts out through the log scores
    loglik[0] = static_cast<double>(pf_mh); loglik[1] = static_cast<double>(pf_thr);
    loglik[2] = static_cast<double>(pf_loop); loglik[3] = static_cast<double>(pf_red);
    loglik[4] = static_cast<double>(pf_rec);
  }
#endif
  if (MODE == 0 && WPB == 8 && !WIDE && !COLLAPSED && a.balance == 1) {   // done: the partner is never "behind" again
    __hip_atomic_store(&k2_prog[threadIdx.x >> 6], 0x7FFFFFFF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_s_setprio(0);
  }
  if (a.M > 0 && live && chain == 0) gibbs_write(static_cast<uint32_t>(a.M - 1));
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
    st->hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID, all 32 bits
#ifdef MISO_K2_WAVETIME
    uint64_t wt_t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wt_t1) : : "memory");
    st->hw_id = static_cast<uint32_t>(wt_t1 - wt_t0);
#endif
  }
}

}  // namespace miso

// kernels_grp.hip -- lane-packed sampler for any isoform count and for paired-end reads.
//
// sampler_wave (kernels.hip) gives one wavefront to one chain; its per-iteration scalar step
// (miso.c:449-552: O(K) transcendentals in one dependency chain) keeps 64 lanes busy for the
// benefit of a single chain.  Here a chain owns G lanes (G = 2..32) and a wavefront carries 64/G
// chains, so that step is shared 64/G ways, exactly as in sampler_k2 -- but for any K <= 32 and for
// the paired-end model:
//   * per-chain vectors (psi, alpha, proposals, scratch, per-isoform constants, counts) live in an
//     LDS slice of the chain; lane `sub` of the chain owns isoforms sub, sub+G, ...: transcendentals
//     run lane-parallel, the reference's left-to-right sums are re-read from LDS by every lane
//     (broadcast reads), so all lanes of a chain hold identical scalars;
//   * Gibbs: the chain's lanes stride over its draw quads (one Philox4x32-10 block = four reads);
//     SE reads carry a u32 compatibility mask, PE reads K u16 fragment indices into the
//     fragment-probability table staged in LDS; picks are counted with LDS atomics on the chain's
//     slice; the PE fragment score is accumulated in 2^-26 fixed point and reduced over the lanes;
//   * single-end Gibbs, class path: the drawing reads are ordered by compatibility class (host.hpp),
//     so a class's pick thresholds are integers computed once per iteration and a read costs one
//     Philox word and (class size - 1) compare+add into register counters -- no per-read LDS
//     traffic, no atomics (class_units below).
// Same arithmetic, same order, same RNG addresses as sampler_wave and the CPU checker.
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "miso_amd.h"
#include "miso_detmath.h"
#include "miso_philox.h"

#pragma clang fp contract(off)

#ifdef MISO_K2_PROFILE
#define GPROF_T(var) const uint64_t var = __builtin_readcyclecounter()
#define GPROF_ADD(acc, t0, t1) acc += (t1) - (t0)
#else
#define GPROF_T(var)
#define GPROF_ADD(acc, t0, t1)
#endif

namespace miso {

namespace {

// LDS accesses of different lanes of one wavefront are ordered by program order once the compiler
// is told not to move them: a wavefront-scope fence is enough (no s_barrier: chains never span waves)
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

__device__ __forceinline__ double seq_sum(const double *v, int n) {
  double acc = 0.0;
  for (int k = 0; k < n; k++) acc = acc + v[k];
  return acc;
}

struct Slice {            // one chain's LDS slice; every array has `ks` entries
  double *psi, *alpha, *psiN, *alphaN, *ta, *tb, *tc, *cst, *isc, *hm1;
  int *cnt, *bas;  // picks of the drawing reads; reads with a single compatible isoform
  // SE class path (only when qstride > 0), per drawing-read class c < MAX_DRAW_CLASSES:
  uint32_t *thr;   //   low words of the cumulative integer thresholds [c x (ks - 1)]
  int *cum;        //   reads of the class whose word is below threshold j [c x (ks - 1)]
  uint32_t *cmask; //   compatibility mask
  int *csize;      //   number of reads
  uint32_t *alw;   //   bit j: threshold j is 2^32 (every word is below it)
  uint32_t *units; //   work units: q | wordmask << 20 | class << 24 [qstride]
  int32_t *stab;   // PE: the event's fixed-point score table [tstride] (when it fits)
};


__device__ __forceinline__ Slice carve(unsigned char *base, int ks, int qs) {
  Slice s;
  double *d = reinterpret_cast<double *>(base);
  s.psi = d; s.alpha = d + ks; s.psiN = d + 2 * ks; s.alphaN = d + 3 * ks; s.ta = d + 4 * ks;
  s.tb = d + 5 * ks; s.tc = d + 6 * ks; s.cst = d + 7 * ks; s.isc = d + 8 * ks; s.hm1 = d + 9 * ks;
  s.cnt = reinterpret_cast<int *>(d + 10 * ks);
  s.bas = s.cnt + ks;
  s.thr = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(d) + 10 * ks * 8 + 2 * ks * 4);
  s.cum = reinterpret_cast<int *>(s.thr + MAX_DRAW_CLASSES * (ks - 1));
  s.cmask = reinterpret_cast<uint32_t *>(s.cum + MAX_DRAW_CLASSES * (ks - 1));
  s.csize = reinterpret_cast<int *>(s.cmask + MAX_DRAW_CLASSES);
  s.alw = reinterpret_cast<uint32_t *>(s.csize + MAX_DRAW_CLASSES);
  s.units = s.alw + MAX_DRAW_CLASSES;
  s.stab = reinterpret_cast<int32_t *>(reinterpret_cast<unsigned char *>(s.thr) + grp_cls_bytes(ks, qs));
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
template <bool LE> __device__ __forceinline__ uint64_t draw_threshold(double c, double T) {
  double est = c / T * 4294967296.0;
  est = (est > 0.0) ? est : 0.0;
  est = (est > 4294967296.0) ? 4294967296.0 : est;
  const int64_t t0 = static_cast<int64_t>(est);
  const int n = thr_pred<LE>(t0 - 1, c, T) + thr_pred<LE>(t0, c, T) + thr_pred<LE>(t0 + 1, c, T);
  int64_t t = t0 - 1 + n;
  if (!thr_pred<LE>(t0 - 2, c, T) || thr_pred<LE>(t0 + 2, c, T)) {  // exact fallback, not taken in practice
    t = t0 < 0 ? 0 : (t0 > 4294967296ll ? 4294967296ll : t0);
    for (int g = 0; g < 4096 && t > 0 && !thr_pred<LE>(t - 1, c, T); g++) t--;
    for (int g = 0; g < 4096 && t < 4294967296ll && thr_pred<LE>(t, c, T); g++) t++;
  }
  return static_cast<uint64_t>(t < 0 ? 0 : t);
}

// The single-end class path's read loop.  The chain's G lanes stride over its work units (the words
// of one Philox block that belong to one class); a lane keeps the thresholds T[] of its current
// class and the counters cj[j] = #{words seen below T[j]} in registers and touches LDS only when
// its class changes.  TW >= the wavefront's largest (class size - 1), compile-time for the unroll.
template <int TW, int G>
__device__ __forceinline__ void class_units(const uint32_t *units, const uint32_t *thr, int *cum, int tw,
                                            int nuw, int n_units, int sub, uint64_t seed,
                                            uint32_t event_id, uint32_t chain, uint32_t iter) {
  uint32_t T[TW]; int cj[TW];
#pragma unroll
  for (int j = 0; j < TW; j++) { T[j] = 0; cj[j] = 0; }
  int cur = -1;
  for (int i0 = 0; i0 < nuw; i0 += G) {
    const int i = i0 + sub;
    const bool active = i < n_units;
    const uint32_t un = active ? units[i] : 0u;
    const uint32_t wm = (un >> 20) & 0xFu;
    const int c = active ? static_cast<int>(un >> 24) : cur;
    const miso_u32x4 u = miso_draw_block(seed, event_id, chain, iter, MISO_SITE_GIBBS, un & 0xFFFFFu);
    if (c != cur) {
      if (cur >= 0) {
#pragma unroll
        for (int j = 0; j < TW; j++) if (j < tw && cj[j]) atomicAdd(&cum[cur * tw + j], cj[j]);
      }
#pragma unroll
      for (int j = 0; j < TW; j++) { T[j] = (j < tw) ? thr[c * tw + j] : 0u; cj[j] = 0; }
      cur = c;
    }
#pragma unroll
    for (int w = 0; w < 4; w++) {
      const uint32_t uw = ((wm >> w) & 1u) ? u.v[w] : 0xFFFFFFFFu;   // never below a 32-bit threshold
#pragma unroll
      for (int j = 0; j < TW; j++) cj[j] += (uw < T[j]) ? 1 : 0;
    }
  }
  if (cur >= 0) {
#pragma unroll
    for (int j = 0; j < TW; j++) if (j < tw && cj[j]) atomicAdd(&cum[cur * tw + j], cj[j]);
  }
}

}  // namespace



template <int G, bool PE>
__global__ __launch_bounds__(256, 2) void sampler_grp(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int CPW = 64 / G;
  const int fp_bytes = PE ? ((a.il * 8 + 15) & ~15) : 0;
  double *lds_fp = reinterpret_cast<double *>(smem);
  if (PE) {
    for (int i = threadIdx.x; i < a.il; i += blockDim.x) lds_fp[i] = a.frag_prob[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / G, sub = lane - grp * G;
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  const long wave_id = static_cast<long>(blockIdx.x) * 4 + wave;
  if (wave_id * CPW >= n_chains) return;  // no block-level barrier below
  long slot = wave_id * CPW + grp;
  const bool live = slot < n_chains;
  if (!live) slot = n_chains - 1;           // shadow a real chain, store nothing
  const int ks = a.kstride;
  const Slice S = carve(smem + fp_bytes + (static_cast<size_t>(wave) * CPW + grp) * grp_slice_bytes(ks, a.qstride, a.tstride), ks, a.qstride);

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
  const uint32_t event_id = a.first_event_id + static_cast<uint32_t>(ev);
  const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
  const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);
  for (int k0 = 0; k0 < Kw; k0 += G) {
    const int k = k0 + sub;
    if (k < K) {
      S.cst[k] = consts[k]; S.isc[k] = consts[K + k]; S.hm1[k] = consts[2 * K + k];
      S.alpha[k] = 0.0; S.psi[k] = 0.0; S.cnt[k] = 0; S.bas[k] = base[k];
    }
  }
  // single-end class path: usable when every chain of the wavefront has a class table
  const int n_dcls = PE ? 0 : E.n_dcls;
  const int n_units = PE ? 0 : E.n_units;
  int ncw = n_dcls, nuw = n_units, tww = PE ? 0 : E.max_cls - 1;
  bool cls_ok = !PE && a.qstride > 0 && (n_dcls > 0 || E.n_draw == 0) && n_units <= a.qstride;
  for (int off = 32; off >= 1; off >>= 1) {
    ncw = max(ncw, __shfl_xor(ncw, off));
    nuw = max(nuw, __shfl_xor(nuw, off));
    tww = max(tww, __shfl_xor(tww, off));
  }
  cls_ok = __all(cls_ok);
  ncw = __builtin_amdgcn_readfirstlane(ncw);
  nuw = __builtin_amdgcn_readfirstlane(nuw);
  tww = __builtin_amdgcn_readfirstlane(tww);
  if (cls_ok) {
    const uint32_t *gm = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_clsmask);
    for (int c0 = 0; c0 < ncw; c0 += G) {
      const int cc = c0 + sub;
      if (cc < n_dcls) { S.cmask[cc] = gm[2 * cc]; S.csize[cc] = static_cast<int>(gm[2 * cc + 1]); }
    }
    // the work units stay in LDS for the whole run: the read loop never waits on global memory
    const uint32_t *gu = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_cls);
    for (int i0 = 0; i0 < nuw; i0 += G) { const int i = i0 + sub; if (i < n_units) S.units[i] = gu[i]; }
  }
  wave_sync();
  Scalars c;
  c.lg_sum = consts[3 * K + 0]; c.lg_each = consts[3 * K + 1]; c.sigma = consts[3 * K + 2];
  c.sd = consts[3 * K + 3]; c.covar = consts[3 * K + 4];

  const uint32_t *masks = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_draw);
  const uint16_t *frags = reinterpret_cast<const uint16_t *>(a.in_pool + E.off_draw);
  const int32_t *sfix = reinterpret_cast<const int32_t *>(a.in_pool + E.off_sfix);
  if (PE && a.tstride > 0) {  // score table into the chain's slice: no per-read gather from L2
    int32_t *stab = S.stab;
    for (int i = sub; i < K * a.il; i += G) stab[i] = sfix[i];
    sfix = stab;
    wave_sync();
  }
  double *samples = reinterpret_cast<double *>(a.out_pool + E.off_samples);
  double *loglik = reinterpret_cast<double *>(a.out_pool + E.off_loglik);
  uint8_t *drawass = a.out_pool + E.off_drawass;
  int32_t *trace = (E.off_trace == NO_TRACE) ? nullptr
                                             : reinterpret_cast<int32_t *>(a.out_pool + E.off_trace);
  const int n_draw = E.n_draw, n_quads = (n_draw + 3) >> 2;
  int64_t rfix = 0; int rbad = 0;

#ifdef MISO_K2_PROFILE
  uint64_t gp_thr = 0, gp_loop = 0, gp_mh = 0;
#endif
  // ---- Gibbs step for the chain's current psi (in S.psi) ----
  auto gibbs = [&](uint32_t iter, bool write_ass) {
    GPROF_T(t0);
    for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K) S.cnt[k] = 0; }
    wave_sync();
    if (!PE && cls_ok && !__any(write_ass)) {
      // integer thresholds of every drawing-read class for the current psi: one class per lane
      const int tw = ks - 1;
      for (int c0 = 0; c0 < ncw; c0 += G) {
        const int cc = c0 + sub;
        if (cc < n_dcls) {
          const uint32_t m = S.cmask[cc];
          const int nv = __popc(m);
          double T = 0.0;
          for (uint32_t mm = m; mm; mm &= mm - 1) T = T + S.psi[__ffs(mm) - 1];
          double cum = 0.0; uint64_t run = 0; uint32_t alw = 0; int j = 0;
          for (uint32_t mm = m; j < nv - 1; mm &= mm - 1, j++) {
            cum = cum + S.psi[__ffs(mm) - 1];
            const uint64_t t = (nv == 2) ? draw_threshold<false>(cum, T) : draw_threshold<true>(cum, T);
            run = t > run ? t : run;          // first j with u < t_j  ==  first j with u < running max
            S.thr[cc * tw + j] = static_cast<uint32_t>(run);
            alw |= static_cast<uint32_t>(run >> 32) << j;
            S.cum[cc * tw + j] = 0;
          }
          for (; j < tw; j++) { S.thr[cc * tw + j] = 0u; S.cum[cc * tw + j] = 0; }
          S.alw[cc] = alw;
        }
      }
      wave_sync();
      GPROF_T(t1);
      GPROF_ADD(gp_thr, t0, t1);
#define MISO_UNITS(TW) class_units<TW, G>(S.units, S.thr, S.cum, tw, nuw, n_units, sub, a.seed, event_id, chain, iter)
      if (tww <= 1) MISO_UNITS(1);
      else if (tww == 2) MISO_UNITS(2);
      else if (tww == 3) MISO_UNITS(3);
      else if (tww == 4) MISO_UNITS(4);
      else if (tww == 5) MISO_UNITS(5);
      else if (tww == 6) MISO_UNITS(6);
      else if (tww == 7) MISO_UNITS(7);
      else if (tww <= 9) MISO_UNITS(9);
      else if (tww <= 12) MISO_UNITS(12);
      else if (tww <= 16) MISO_UNITS(16);
      else if (tww <= 23) MISO_UNITS(23);
      else MISO_UNITS(31);
#undef MISO_UNITS
      wave_sync();
      // cumulative counts -> picks per isoform: member j of the class got C_j - C_(j-1) reads
      for (int c0 = 0; c0 < ncw; c0 += G) {
        const int cc = c0 + sub;
        if (cc < n_dcls) {
          const uint32_t m = S.cmask[cc], alw = S.alw[cc];
          const int nv = __popc(m), nc = S.csize[cc];
          int prev = 0, j = 0;
          for (uint32_t mm = m; mm; mm &= mm - 1, j++) {
            const int Cj = (j == nv - 1 || ((alw >> j) & 1u)) ? nc : S.cum[cc * tw + j];
            if (Cj != prev) atomicAdd(&S.cnt[__ffs(mm) - 1], Cj - prev);
            prev = Cj;
          }
        }
      }
      wave_sync();
      GPROF_T(t2);
      GPROF_ADD(gp_loop, t1, t2);
      return;
    }
    int64_t acc = 0; int bad = 0;
    const bool small = Kw <= 8;
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
        if (small) {   // K <= 8: exact unroll for the wavefront's isoform count
          const uint16_t *row = frags + static_cast<size_t>(r) * K;
          switch (Kw) {
          case 2: sel = pick_direct<2, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel); break;
          case 3: sel = pick_direct<3, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel); break;
          case 4: sel = pick_direct<4, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel); break;
          case 5: sel = pick_direct<5, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel); break;
          case 6: sel = pick_direct<6, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel); break;
          case 7: sel = pick_direct<7, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel); break;
          default: sel = pick_direct<8, PE>(ps, m4[j], row, K, lds_fp, on, u.v[j], fsel); break;
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

  // ---- alpha' = alpha + sd z ; psi' = logit_inv(alpha')  (miso.c:449-471) ----
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
        S.ta[k] = miso_det_exp(an);
      }
    }
    wave_sync();
    const double sumexp = seq_sum(S.ta, K - 1) + 1.0;
    for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K - 1) psi_out[k] = S.ta[k] / sumexp; }
    wave_sync();
    const double sumpsi = seq_sum(psi_out, K - 1);
    if (sub == 0) psi_out[K - 1] = 1 - sumpsi;
    wave_sync();
  };

  // ---- joint log score of x under the current counts (miso.c:243-307; PE miso_paired.c:133-174) ----
  auto joint = [&](double *x, double readProbPE) {
    for (int k0 = 0; k0 < Kw; k0 += G) {
      const int k = k0 + sub;
      if (k < K) { const double lx = miso_det_log(x[k]); S.ta[k] = lx; S.tb[k] = lx + S.cst[k]; }
    }
    wave_sync();
    double maxv = S.tb[0];
    for (int k = 1; k < K; k++) { const double v = S.tb[k]; if (v > maxv) maxv = v; }
    for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K) S.tc[k] = miso_det_exp(S.tb[k] - maxv); }
    wave_sync();
    const double lse = miso_det_log(seq_sum(S.tc, K)) + maxv;
    double readProb = 0.0, assProb = 0.0, psiProb = 0.0;
    for (int k = 0; k < K; k++) {
      const int ck = count_of(k);
      if (ck != 0) {
        if (!PE) readProb = readProb + static_cast<double>(ck) * S.isc[k];
        assProb = assProb + static_cast<double>(ck) * (S.tb[k] - lse);
      }
    }
    if (PE) readProb = readProbPE;
    for (int k = 0; k < K; k++) psiProb = psiProb + S.hm1[k] * S.ta[k];
    psiProb = psiProb + c.lg_sum;
    psiProb = psiProb - c.lg_each;
    wave_sync();  // scratch is reused by the next call
    return readProb + assProb + psiProb;
  };

  // ---- log density of theta under the logistic normal centred at mu (miso.c:97-122) ----
  auto prop_score = [&](double *theta, double *mu) {
    double ltheta = 1.0, prod = 1.0;
    for (int i = 0; i < K - 1; i++) { const double t = theta[i]; ltheta = ltheta - t; prod = prod * t; }
    prod = 1.0 / prod / ltheta;
    for (int k0 = 0; k0 < Kw; k0 += G) {
      const int k = k0 + sub;
      if (k < K - 1) {
        const double tmp = miso_det_log(theta[k] / ltheta) - mu[k];
        S.ta[k] = (-0.5) * tmp * tmp / c.sigma;
      }
    }
    wave_sync();
    const double expPart = seq_sum(S.ta, K - 1);
    wave_sync();
    return miso_det_log(c.covar * prod * miso_det_exp(expPart));
  };

  // ---- initial state: miso.c:330-447 (START_AUTO / START_UNIFORM), then miso.c:834, 841 ----
  if (a.start == MISO_START_AUTO && K != 2)
    for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K - 1) S.alpha[k] = 1.0 / (K - 1); }
  wave_sync();
  uint32_t accept_word = 0;
  propose(S.alpha, S.alpha, S.psi, MISO_ITER_INIT, accept_word);
  gibbs(MISO_ITER_INIT, live && chain == 0 && a.M == 0);

  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;

  for (int m = 0; m < a.M; m++) {
    for (int k = 0; k < K; k++) hash = (hash ^ static_cast<uint32_t>(count_of(k))) * 0x100000001B3ull;
    if (trace && live)
      for (int k0 = 0; k0 < Kw; k0 += G) {
        const int k = k0 + sub;
        if (k < K) trace[(static_cast<size_t>(m) * a.C + chain) * K + k] = count_of(k);
      }
    GPROF_T(m0);
    propose(S.alpha, S.alphaN, S.psiN, static_cast<uint32_t>(m), accept_word);
    const double rp = PE ? (rbad ? miso_u2d(0x7FF8000000000000ull)
                                 : static_cast<double>(rfix) * (1.0 / MISO_SFIX_SCALE))
                         : 0.0;
    const double pp = joint(S.psiN, rp);
    const double pc = joint(S.psi, rp);
    const double ptoCS = prop_score(S.psi, S.alphaN);
    const double ctoPS = prop_score(S.psiN, S.alpha);
    const double acceptP = (m > 0) ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    double cJS = pc;
    if (acc) {
      for (int k0 = 0; k0 < Kw; k0 += G) {
        const int k = k0 + sub;
        if (k < K) S.psi[k] = S.psiN[k];
        if (k < K - 1) S.alpha[k] = S.alphaN[k];
      }
      cJS = pp; accepted++;
    }
    wave_sync();
    GPROF_T(m1);
    GPROF_ADD(gp_mh, m0, m1);
    if (m >= a.B) {  // miso.c:882-893
      if (lagCounter == a.lag - 1) {
        if (live) {
          const size_t col = static_cast<size_t>(noS) + chain;
          for (int k0 = 0; k0 < Kw; k0 += G) { const int k = k0 + sub; if (k < K) samples[col * K + k] = S.psi[k]; }
          if (sub == 0) loglik[col] = cJS;
        }
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    gibbs(static_cast<uint32_t>(m), live && chain == 0 && m == a.M - 1);
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
      st->counts_hash = hash; st->accepted = accepted; st->pad = 0;
    }
  }
}

#define MISO_INSTANTIATE_GRP(G) \
  template __global__ void sampler_grp<G, false>(const KernelArgs); \
  template __global__ void sampler_grp<G, true>(const KernelArgs);
MISO_INSTANTIATE_GRP(2)
MISO_INSTANTIATE_GRP(4)
MISO_INSTANTIATE_GRP(8)
MISO_INSTANTIATE_GRP(16)
MISO_INSTANTIATE_GRP(32)

}  // namespace miso

// kernels.hip -- CDNA4 (gfx950) kernels of the MISO posterior sampler.
//
// What one chain does per iteration (reference: miso.c:845-900, miso_paired.c:451-498):
//   1. propose psi' by a Gaussian step in logit space          miso.c:449-471
//   2. Metropolis-Hastings accept/reject using the joint score  miso.c:493-552, 869-880
//      of the CURRENT read->isoform assignment                  miso.c:243-307 (PE: miso_paired.c:133-174)
//   3. record psi every `lag` iterations after burn-in          miso.c:882-893
//   4. Gibbs: re-draw every read's isoform from a categorical   miso.c:30-91 (PE: miso_paired.c:24-86)
// Steps 1-2 are O(K) transcendental math, step 4 is O(reads) and is the hot loop.
//
// Mapping used by `sampler_wave` (one wavefront = one (event, chain)):
//   * lane k (< K <= 64) owns isoform k: psi_k, alpha_k, log psi_k, count_k.  Transcendentals run
//     lane-parallel; the reference's left-to-right sums are reproduced with uniform-lane
//     broadcasts (v_readlane), so every lane holds the same, reference-ordered total.
//   * in the Gibbs step lane l owns draw quads l, l+64, ...: one Philox4x32 block = the four
//     uniforms of four consecutive drawing reads (include/miso_philox.h), whose packed
//     compatibility masks (SE) / fragment indices (PE) it loads as one coalesced vector.
//   * the per-isoform counts come back through wave ballots (K <= 4) or LDS atomics.
// No MFMA: there is no contraction here; the work is integer hashing, compares and a little
// f64.  All floating point follows include/miso_detmath.h (no implicit FMA) so the CPU checker
// reproduces every bit.
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "miso_amd.h"
#include "miso_detmath.h"
#include "miso_philox.h"

#pragma clang fp contract(off)

namespace miso {

namespace {

__device__ __forceinline__ double bcast(double x, int lane) {
  // lane is wave-uniform: two v_readlane_b32
  const uint64_t u = miso_d2u(x);
  const uint32_t lo = __builtin_amdgcn_readlane(static_cast<int>(u), lane);
  const uint32_t hi = __builtin_amdgcn_readlane(static_cast<int>(u >> 32), lane);
  return miso_u2d((static_cast<uint64_t>(hi) << 32) | lo);
}

__device__ __forceinline__ int bcast_i(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }

// left-to-right sum of lanes 0..n-1, starting from 0.0 (the reference's accumulation order)
__device__ __forceinline__ double seq_sum(double x, int n) {
  double acc = 0.0;
  for (int k = 0; k < n; k++) acc = acc + bcast(x, k);
  return acc;
}

struct ChainConsts {
  double cst, iscore, hm1;               // lane k: per-isoform constants
  double lg_sum, lg_each, sigma, sd, covar;
};

// log of the joint score given per-isoform counts: miso.c:243-307 with the per-read sums of
// miso.c:267-271 and 152-156 taken as sum_k count_k * value_k (counter-mode contract).
template <bool PE>
__device__ __forceinline__ double joint_score(double x, int cnt, double readProbPE,
                                              const ChainConsts &c, int K, int lane) {
  const double lx = (lane < K) ? miso_det_log(x) : 0.0;
  const double lp = lx + c.cst;
  double maxv = bcast(lp, 0);
  for (int k = 1; k < K; k++) { const double v = bcast(lp, k); if (v > maxv) maxv = v; }
  const double ex = (lane < K) ? miso_det_exp(lp - maxv) : 0.0;
  const double lse = miso_det_log(seq_sum(ex, K)) + maxv;
  const double lpn = lp - lse;
  double readProb = 0.0, assProb = 0.0, psiProb = 0.0;
  for (int k = 0; k < K; k++) {
    const int ck = bcast_i(cnt, k);
    if (ck != 0) {
      if (!PE) readProb = readProb + static_cast<double>(ck) * bcast(c.iscore, k);
      assProb = assProb + static_cast<double>(ck) * bcast(lpn, k);
    }
  }
  if (PE) readProb = readProbPE;
  for (int k = 0; k < K; k++) psiProb = psiProb + bcast(c.hm1, k) * bcast(lx, k);
  psiProb = psiProb + c.lg_sum;
  psiProb = psiProb - c.lg_each;
  return readProb + assProb + psiProb;
}

// log density of the logistic-normal proposal: miso.c:97-122 (theta, mu on lanes 0..K-2)
__device__ __forceinline__ double proposal_score(double theta, double mu, const ChainConsts &c,
                                                 int K, int lane) {
  double ltheta = 1.0, prod = 1.0;
  for (int i = 0; i < K - 1; i++) { const double t = bcast(theta, i); ltheta = ltheta - t; prod = prod * t; }
  prod = 1.0 / prod / ltheta;
  double term = 0.0;
  if (lane < K - 1) {
    const double tmp = miso_det_log(theta / ltheta) - mu;
    term = (-0.5) * tmp * tmp / c.sigma;
  }
  const double expPart = seq_sum(term, K - 1);
  const double pdf = c.covar * prod * miso_det_exp(expPart);
  return miso_det_log(pdf);
}

// alpha' = alpha + sd * N(0,1), psi' = logit_inv(alpha'): miso.c:449-471, 184-241
__device__ __forceinline__ void propose(double alpha, double &alphaN, double &psiN, uint32_t &accept_word,
                                        const ChainConsts &c, int K, int lane, uint64_t seed,
                                        uint32_t event_id, uint32_t chain, uint32_t iter) {
  double z = 0.0;
  const int w = 2 + 2 * lane;  // words 2+2j, 3+2j of the MH word stream feed normal j
  miso_u32x4 b = miso_draw_block(seed, event_id, chain, iter, MISO_SITE_MH,
                                 static_cast<uint32_t>(w >> 2));
  if (lane < K - 1) z = miso_det_norm_from_unif(miso_u01(b.v[w & 3]), miso_u01(b.v[(w & 3) + 1]));
  accept_word = __builtin_amdgcn_readlane(static_cast<int>(b.v[0]), 0);  // block 0, word 0
  alphaN = alpha + c.sd * z;
  const double e = (lane < K - 1) ? miso_det_exp(alphaN) : 0.0;
  const double sumexp = seq_sum(e, K - 1) + 1.0;
  const double p = e / sumexp;
  const double sumpsi = seq_sum(p, K - 1);
  psiN = (lane == K - 1) ? (1 - sumpsi) : p;
}

}  // namespace

// One wavefront per (event, chain); 4 wavefronts per workgroup.
template <bool PE>
__global__ __launch_bounds__(256) void sampler_wave(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: [0, il*8) fragment probabilities (PE); then 4 x 64 ints of count scratch
  double *lds_fp = reinterpret_cast<double *>(smem);
  const int fp_bytes = PE ? ((a.il * 8 + 15) & ~15) : 0;
  int *lds_cnt_all = reinterpret_cast<int *>(smem + fp_bytes);
  if (PE) {
    for (int i = threadIdx.x; i < a.il; i += blockDim.x) lds_fp[i] = a.frag_prob[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long slot = static_cast<long>(blockIdx.x) * 4 + wave;
  if (slot >= static_cast<long>(a.n_slots) * a.C) return;  // no block barrier below this line
  int *lds_cnt = lds_cnt_all + wave * 64;

  const int ev = a.slot_event[slot / a.C];
  const uint32_t chain = static_cast<uint32_t>(slot % a.C);
  const DevEvent E = a.events[ev];
  const int K = E.K;
  const uint32_t event_id = E.has_id ? E.explicit_id : a.first_event_id + static_cast<uint32_t>(ev);
  const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
  const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);

  ChainConsts c;
  c.cst = (lane < K) ? consts[lane] : 0.0;
  c.iscore = (lane < K) ? consts[K + lane] : 0.0;
  c.hm1 = (lane < K) ? consts[2 * K + lane] : 0.0;
  c.lg_sum = consts[3 * K + 0]; c.lg_each = consts[3 * K + 1]; c.sigma = consts[3 * K + 2];
  c.sd = consts[3 * K + 3]; c.covar = consts[3 * K + 4];
  const int base_cnt = (lane < K) ? base[lane] : 0;

  const uint32_t *masks = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_draw);
  // from 33 isoforms on a mask has two words: the high words follow the (quad-padded) low words (runtime.hip upload)
  const uint32_t *masks_hi = masks + ((static_cast<size_t>(E.n_draw) + 3) & ~static_cast<size_t>(3));
  const bool wide_masks = K > 32;
  const uint16_t *frags = reinterpret_cast<const uint16_t *>(a.in_pool + E.off_draw);
  const int32_t *sfix = reinterpret_cast<const int32_t *>(a.in_pool + E.off_sfix);
  double *samples = reinterpret_cast<double *>(a.out_pool + E.off_samples);
  double *loglik = reinterpret_cast<double *>(a.out_pool + E.off_loglik);
  uint8_t *drawass = a.out_pool + E.off_drawass;
  int32_t *trace = (E.off_trace == NO_TRACE) ? nullptr
                                             : reinterpret_cast<int32_t *>(a.out_pool + E.off_trace);
  const int n_draw = E.n_draw, n_quads = (n_draw + 3) >> 2;

  double psi = 0.0, alpha = 0.0;  // lane k: psi_k; lane i < K-1: alpha_i
  int cnt = 0;                    // lane k: reads currently assigned to isoform k
  int64_t rfix = 0;               // PE: fixed-point sum of the assigned reads' fragment scores
  int rbad = 0;

  // Gibbs step for the current psi; iteration `iter` addresses the uniforms.
  auto gibbs = [&](uint32_t iter, bool write_ass) {
    int my_cnt = 0;  // lane k accumulates count_k (ballot path)
    int64_t acc = 0; int bad = 0;
    const bool use_lds = K > 4;
    if (use_lds) { lds_cnt[lane] = 0; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
    for (int q0 = 0; q0 < n_quads; q0 += 64) {
      const int q = q0 + lane;
      const bool active = q < n_quads;
      miso_u32x4 u = miso_draw_block(a.seed, event_id, chain, iter, MISO_SITE_GIBBS,
                                     static_cast<uint32_t>(q));
      uint64_t m4[4] = {0, 0, 0, 0};
      if (!PE && active) {
        const uint4 v = *reinterpret_cast<const uint4 *>(masks + 4 * static_cast<size_t>(q));
        m4[0] = v.x; m4[1] = v.y; m4[2] = v.z; m4[3] = v.w;
        if (wide_masks) {
          const uint4 h = *reinterpret_cast<const uint4 *>(masks_hi + 4 * static_cast<size_t>(q));
          m4[0] |= static_cast<uint64_t>(h.x) << 32; m4[1] |= static_cast<uint64_t>(h.y) << 32;
          m4[2] |= static_cast<uint64_t>(h.z) << 32; m4[3] |= static_cast<uint64_t>(h.w) << 32;
        }
      }
      int sel[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = 4 * q + j;
        const bool live = active && r < n_draw;
        sel[j] = -1;
        // pass 1: total weight of the compatible isoforms, ascending k (miso.c:11-22)
        double T = 0.0; int nv = 0;
        for (int k = 0; k < K; k++) {
          const double pk = bcast(psi, k);
          if (PE) {
            const uint16_t f = live ? frags[static_cast<size_t>(r) * K + k] : FRAG_NONE;
            if (f != FRAG_NONE) { T = T + pk * lds_fp[f]; nv++; }
          } else if (live && ((m4[j] >> k) & 1ull)) { T = T + pk; nv++; }
        }
        const double rnd = miso_u01(u.v[j]) * T;
        // pass 2: first valid isoform whose cumulative weight stops the scan (miso.c:69-80)
        double cum = 0.0; int idx = 0, lastv = -1; uint16_t fsel = 0;
        for (int k = 0; k < K; k++) {
          const double pk = bcast(psi, k);
          bool valid; uint16_t f = 0;
          if (PE) {
            f = live ? frags[static_cast<size_t>(r) * K + k] : FRAG_NONE;
            valid = f != FRAG_NONE;
            if (valid) cum = cum + pk * lds_fp[f];
          } else {
            valid = live && ((m4[j] >> k) & 1ull);
            if (valid) cum = cum + pk;
          }
          if (valid) {
            const bool stop = (nv == 2) ? (idx == 0 ? (rnd < cum) : true) : !(rnd > cum);
            if (sel[j] < 0 && stop) { sel[j] = k; fsel = f; }
            lastv = k; idx++;
            if (sel[j] < 0 && idx == nv) { sel[j] = lastv; fsel = f; }
          }
        }
        if (PE && sel[j] >= 0) {
          const int32_t v = sfix[static_cast<size_t>(sel[j]) * a.il + fsel];
          if (v == SFIX_BAD) bad = 1; else acc += v;
        }
        if (write_ass && sel[j] >= 0) drawass[r] = static_cast<uint8_t>(sel[j]);
      }
      if (use_lds) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (sel[j] >= 0) atomicAdd(&lds_cnt[sel[j]], 1);
      } else {
        for (int k = 0; k < K; k++) {
          int n = 0;
#pragma unroll
          for (int j = 0; j < 4; j++) n += __popcll(__ballot(sel[j] == k));
          if (lane == k) my_cnt += n;
        }
      }
    }
    if (use_lds) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      my_cnt = lds_cnt[lane];
    }
    cnt = base_cnt + ((lane < K) ? my_cnt : 0);
    if (PE) {
      for (int off = 32; off > 0; off >>= 1) {
        acc += __shfl_xor(acc, off);
        bad |= __shfl_xor(bad, off);
      }
      rfix = E.base_sfix + acc;
      rbad = bad | E.base_bad;
    }
  };

  // ---- initial state: miso.c:330-447 (START_AUTO / START_UNIFORM), then miso.c:834, 841 ----
  if (a.start == MISO_START_AUTO && K != 2) alpha = (lane < K - 1) ? 1.0 / (K - 1) : 0.0;
  uint32_t accept_word = 0;
  {
    double aN, pN;
    propose(alpha, aN, pN, accept_word, c, K, lane, a.seed, event_id, chain, MISO_ITER_INIT);
    alpha = aN; psi = pN;
  }
  gibbs(MISO_ITER_INIT, chain == 0 && a.M == 0);

  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;

  RoundOpen ro(a);
  for (int m = 0; m < a.M; m++) {
    const bool opens = ro.at(a, m);   // a round's first iteration: no proposal terms (miso.c:866)
    for (int k = 0; k < K; k++) hash = (hash ^ static_cast<uint32_t>(bcast_i(cnt, k))) * 0x100000001B3ull;
    if (trace && lane < K) trace[(static_cast<size_t>(m) * a.C + chain) * K + lane] = cnt;

    double alphaN, psiN;
    propose(alpha, alphaN, psiN, accept_word, c, K, lane, a.seed, event_id, chain,
            static_cast<uint32_t>(m));
    const double rp = PE ? (rbad ? miso_u2d(0x7FF8000000000000ull)
                                 : static_cast<double>(rfix) * (1.0 / MISO_SFIX_SCALE))
                         : 0.0;
    const double pp = joint_score<PE>(psiN, cnt, rp, c, K, lane);
    const double pc = joint_score<PE>(psi, cnt, rp, c, K, lane);
    const double ptoCS = proposal_score(psi, alphaN, c, K, lane);
    const double ctoPS = proposal_score(psiN, alpha, c, K, lane);
    const double acceptP = !opens ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    double cJS = pc;
    if (acc) { psi = psiN; alpha = alphaN; cJS = pp; accepted++; }

    if (m >= a.B) {  // miso.c:882-893
      if (lagCounter == a.lag - 1) {
        const size_t col = static_cast<size_t>(noS) + chain;
        if (lane < K) samples[col * K + lane] = psi;
        if (lane == 0) loglik[col] = cJS;
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    gibbs(static_cast<uint32_t>(m), chain == 0 && m == a.M - 1);
  }
  for (int k = 0; k < K; k++) hash = (hash ^ static_cast<uint32_t>(bcast_i(cnt, k))) * 0x100000001B3ull;
  if (trace && lane < K) trace[(static_cast<size_t>(a.M) * a.C + chain) * K + lane] = cnt;
  if (lane == 0) {
    ChainStats *st = reinterpret_cast<ChainStats *>(a.out_pool + E.off_stats) + chain;
    st->counts_hash = hash;
    st->accepted = accepted;
    st->hw_id = 0;
  }
}

template __global__ void sampler_wave<false>(const KernelArgs);
template __global__ void sampler_wave<true>(const KernelArgs);

// ---- arithmetic-contract self tests (tests/test_gpu_contract.py) ----
__global__ void selftest_detmath_kernel(const double *x, int n, double *e, double *l, double *s,
                                        double *q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  e[i] = miso_det_exp(x[i]);
  l[i] = miso_det_log(x[i]);
  s[i] = miso_det_sqrt(x[i]);
  q[i] = miso_det_qnorm(x[i]);
}

__global__ void selftest_philox_kernel(const uint32_t *in6, int n, uint32_t *out4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t *p = in6 + 6 * static_cast<size_t>(i);
  const miso_u32x4 o = miso_philox4x32(p[0], p[1], p[2], p[3], p[4], p[5]);
  for (int j = 0; j < 4; j++) out4[4 * static_cast<size_t>(i) + j] = o.v[j];
}

}  // namespace miso

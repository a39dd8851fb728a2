// kernels_big.hip -- sampler_big: genes of 65 ... MISO_MAX_ISOFORMS isoforms (round 5).
//
// The reference has no limit on a gene's isoforms (miso.c:696, gff.c:684); rounds 1 - 4 stopped at 64 (sampler_wave: lane k =
// isoform k, kernels.hip).  Such genes are a handful per annotation, so this kernel is written for being RIGHT, not fast:
// one wavefront per chain, the chain's vectors (psi, alpha, their proposals, logs, scratch, the per-isoform constants and
// counts) in LDS, every transcendental lane-parallel over the isoforms (k = lane, lane + 64, ...), every left-to-right sum of
// the reference (miso.c:97-163, 243-307, 449-552) walked by all lanes over LDS -- the same values in the same order as
// sampler_wave and the CPU checker's counter mode; a read's compatibility mask has (K + 31) / 32 words, one plane of whole
// quads per word (runtime.hip upload).  Same RNG addresses (include/miso_philox.h): bit-exact against the checker
// (tests/test_gpu_parity.py::test_more_than_sixty_four_isoforms_bit_exact).
#include <hip/hip_runtime.h>

#include "device.hpp"
#include "miso_amd.h"
#include "miso_detmath.h"
#include "miso_philox.h"

#pragma clang fp contract(off)

namespace miso {

namespace {

// LDS traffic between lanes of ONE wavefront: program order once the compiler may not move the accesses
__device__ __forceinline__ void bsync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

struct BigSlice {   // one chain's vectors, each `ks` entries
  double *psi, *alpha, *psiN, *alphaN, *lx, *lp, *t0, *t1, *cst, *isc, *hm1;
  int *cnt, *base;
};
constexpr int BIG_DOUBLES = 11;

// left-to-right sum of v[0 .. n-1] starting from 0.0 (the reference's accumulation order; every lane the same walk)
__device__ __forceinline__ double big_sum(const double *v, int n) {
  double acc = 0.0;
  for (int k = 0; k < n; k++) acc = acc + v[k];
  return acc;
}

struct BigConsts { double lg_sum, lg_each, sigma, sd, covar; };

// miso.c:243-307 with the per-read sums taken as sum_k count_k value_k (counter contract); x: the psi vector scored
template <bool PE>
__device__ __forceinline__ double big_joint(const BigSlice &S, const double *x, double readProbPE, const BigConsts &c, int K, int lane) {
  for (int k = lane; k < K; k += 64) { const double l = miso_det_log(x[k]); S.lx[k] = l; S.lp[k] = l + S.cst[k]; }
  bsync();
  double maxv = S.lp[0];
  for (int k = 1; k < K; k++) { const double v = S.lp[k]; if (v > maxv) maxv = v; }
  for (int k = lane; k < K; k += 64) S.t0[k] = miso_det_exp(S.lp[k] - maxv);
  bsync();
  const double lse = miso_det_log(big_sum(S.t0, K)) + maxv;
  double readProb = 0.0, assProb = 0.0, psiProb = 0.0;
  for (int k = 0; k < K; k++) {
    const int ck = S.cnt[k];
    if (ck != 0) {
      if (!PE) readProb = readProb + static_cast<double>(ck) * S.isc[k];
      assProb = assProb + static_cast<double>(ck) * (S.lp[k] - lse);
    }
  }
  if (PE) readProb = readProbPE;
  for (int k = 0; k < K; k++) psiProb = psiProb + S.hm1[k] * S.lx[k];
  psiProb = psiProb + c.lg_sum;
  psiProb = psiProb - c.lg_each;
  bsync();   // (lx, lp, t0 are reused by the next call)
  return readProb + assProb + psiProb;
}

// miso.c:97-122: log density of the logistic-normal proposal, theta and mu on entries 0 .. K-2
__device__ __forceinline__ double big_proposal(const BigSlice &S, const double *theta, const double *mu, const BigConsts &c, int K, int lane) {
  double ltheta = 1.0, prod = 1.0;
  for (int i = 0; i < K - 1; i++) { const double t = theta[i]; ltheta = ltheta - t; prod = prod * t; }
  prod = 1.0 / prod / ltheta;
  for (int i = lane; i < K - 1; i += 64) {
    const double tmp = miso_det_log(theta[i] / ltheta) - mu[i];
    S.t1[i] = (-0.5) * tmp * tmp / c.sigma;
  }
  bsync();
  const double expPart = big_sum(S.t1, K - 1);
  const double pdf = c.covar * prod * miso_det_exp(expPart);
  bsync();
  return miso_det_log(pdf);
}

}  // namespace

// One wavefront (= one workgroup of 64 threads) per (event, chain).
template <bool PE>
__global__ __launch_bounds__(64) void sampler_big(const KernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_big[];
  double *lds_fp = reinterpret_cast<double *>(smem_big);
  const int fp_bytes = PE ? ((a.il * 8 + 15) & ~15) : 0;
  const int lane = threadIdx.x;
  if (PE) {
    for (int i = lane; i < a.il; i += 64) lds_fp[i] = a.frag_prob[i];
    bsync();
  }
  const long slot = blockIdx.x;
  if (slot >= static_cast<long>(a.n_slots) * a.C) return;
  const int ks = a.kstride;
  BigSlice S;
  {
    double *d = reinterpret_cast<double *>(smem_big + fp_bytes);
    S.psi = d; S.alpha = d + ks; S.psiN = d + 2 * ks; S.alphaN = d + 3 * ks; S.lx = d + 4 * ks; S.lp = d + 5 * ks;
    S.t0 = d + 6 * ks; S.t1 = d + 7 * ks; S.cst = d + 8 * ks; S.isc = d + 9 * ks; S.hm1 = d + 10 * ks;
    S.cnt = reinterpret_cast<int *>(d + BIG_DOUBLES * ks); S.base = S.cnt + ks;
  }
  const int ev = a.slot_event[slot / a.C];
  const uint32_t chain = static_cast<uint32_t>(slot % a.C);
  const DevEvent E = a.events[ev];
  const int K = E.K;
  const uint32_t event_id = E.has_id ? E.explicit_id : a.first_event_id + static_cast<uint32_t>(ev);
  const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
  const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);
  for (int k = lane; k < K; k += 64) {
    S.cst[k] = consts[k]; S.isc[k] = consts[K + k]; S.hm1[k] = consts[2 * K + k];
    S.base[k] = base[k]; S.cnt[k] = 0; S.psi[k] = 0.0; S.alpha[k] = 0.0;
  }
  BigConsts c;
  c.lg_sum = consts[3 * K + 0]; c.lg_each = consts[3 * K + 1]; c.sigma = consts[3 * K + 2];
  c.sd = consts[3 * K + 3]; c.covar = consts[3 * K + 4];
  bsync();

  const int n_draw = E.n_draw, n_quads = (n_draw + 3) >> 2;
  const int W32 = (K + 31) >> 5;
  const uint32_t *masks = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_draw);   // W32 planes of 4 n_quads words
  const size_t plane = static_cast<size_t>(n_quads) * 4;
  const uint16_t *frags = reinterpret_cast<const uint16_t *>(a.in_pool + E.off_draw);
  const int32_t *sfix = reinterpret_cast<const int32_t *>(a.in_pool + E.off_sfix);
  double *samples = reinterpret_cast<double *>(a.out_pool + E.off_samples);
  double *loglik = reinterpret_cast<double *>(a.out_pool + E.off_loglik);
  uint8_t *drawass = a.out_pool + E.off_drawass;
  int32_t *trace = (E.off_trace == NO_TRACE) ? nullptr : reinterpret_cast<int32_t *>(a.out_pool + E.off_trace);
  int64_t rfix = 0; int rbad = 0;

  // Gibbs step for the current psi (miso.c:30-91; paired-end miso_paired.c:24-86): lane l owns the reads l, l + 64, ...
  // of every block of 64 quads exactly as sampler_wave does (quad q = its reads 4q .. 4q+3, one Philox block)
  auto gibbs = [&](uint32_t iter, bool write_ass) {
    for (int k = lane; k < K; k += 64) S.cnt[k] = 0;
    bsync();
    int64_t acc = 0; int bad = 0;
    for (int q0 = 0; q0 < n_quads; q0 += 64) {
      const int q = q0 + lane;
      const bool active = q < n_quads;
      const miso_u32x4 u = miso_draw_block(a.seed, event_id, chain, iter, MISO_SITE_GIBBS, static_cast<uint32_t>(q));
      for (int j = 0; j < 4; j++) {
        const int r = 4 * q + j;
        const bool live = active && r < n_draw;
        if (!live) continue;
        // pass 1: total weight of the compatible isoforms, ascending k (miso.c:11-22)
        double T = 0.0; int nv = 0;
        if (PE) {
          for (int k = 0; k < K; k++) {
            const uint16_t f = frags[static_cast<size_t>(r) * K + k];
            if (f != FRAG_NONE) { T = T + S.psi[k] * lds_fp[f]; nv++; }
          }
        } else {
          for (int w = 0; w < W32; w++) {
            uint32_t m = masks[plane * w + r];
            while (m) { const int k = 32 * w + __builtin_ctz(m); m &= m - 1; T = T + S.psi[k]; nv++; }
          }
        }
        const double rnd = miso_u01(u.v[j]) * T;
        // pass 2: first valid isoform whose cumulative weight stops the scan (miso.c:69-80)
        double cum = 0.0; int idx = 0, sel = -1; uint16_t fsel = 0;
        if (PE) {
          for (int k = 0; k < K && sel < 0; k++) {
            const uint16_t f = frags[static_cast<size_t>(r) * K + k];
            if (f == FRAG_NONE) continue;
            cum = cum + S.psi[k] * lds_fp[f];
            const bool stop = (nv == 2) ? (idx == 0 ? (rnd < cum) : true) : !(rnd > cum);
            idx++;
            if (stop || idx == nv) { sel = k; fsel = f; }
          }
        } else {
          for (int w = 0; w < W32 && sel < 0; w++) {
            uint32_t m = masks[plane * w + r];
            while (m && sel < 0) {
              const int k = 32 * w + __builtin_ctz(m); m &= m - 1;
              cum = cum + S.psi[k];
              const bool stop = (nv == 2) ? (idx == 0 ? (rnd < cum) : true) : !(rnd > cum);
              idx++;
              if (stop || idx == nv) sel = k;
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
    bsync();
    for (int k = lane; k < K; k += 64) S.cnt[k] += S.base[k];
    if (PE) {
      for (int off = 32; off > 0; off >>= 1) { acc += __shfl_xor(acc, off); bad |= __shfl_xor(bad, off); }
      rfix = E.base_sfix + acc;
      rbad = bad | E.base_bad;
    }
    bsync();
  };
  // alpha' = alpha + sd N(0,1), psi' = logit_inv(alpha') (miso.c:449-471, 184-241); normal j from words 2 + 2j, 3 + 2j of the
  // iteration's MH word stream, the accept word = word 0
  auto propose = [&](const double *al, double *alN, double *psN, uint32_t iter, uint32_t &accept_word) {
    for (int j = lane; j < K - 1; j += 64) {
      const int w = 2 + 2 * j;
      const miso_u32x4 b = miso_draw_block(a.seed, event_id, chain, iter, MISO_SITE_MH, static_cast<uint32_t>(w >> 2));
      const double z = miso_det_norm_from_unif(miso_u01(b.v[w & 3]), miso_u01(b.v[(w & 3) + 1]));
      const double aN = al[j] + c.sd * z;
      alN[j] = aN;
      S.t0[j] = miso_det_exp(aN);
    }
    accept_word = miso_draw_block(a.seed, event_id, chain, iter, MISO_SITE_MH, 0u).v[0];
    bsync();
    const double sumexp = big_sum(S.t0, K - 1) + 1.0;
    for (int j = lane; j < K - 1; j += 64) psN[j] = S.t0[j] / sumexp;
    bsync();
    const double sumpsi = big_sum(psN, K - 1);
    if (lane == 0) { psN[K - 1] = 1 - sumpsi; alN[K - 1] = 0.0; }
    bsync();
  };

  // ---- initial state: miso.c:330-447 (START_AUTO / START_UNIFORM), then miso.c:834, 841 ----
  if (a.start == MISO_START_AUTO) for (int j = lane; j < K - 1; j += 64) S.alpha[j] = 1.0 / (K - 1);
  bsync();
  uint32_t accept_word = 0;
  propose(S.alpha, S.alphaN, S.psiN, MISO_ITER_INIT, accept_word);
  for (int k = lane; k < K; k += 64) { S.alpha[k] = S.alphaN[k]; S.psi[k] = S.psiN[k]; }
  bsync();
  gibbs(MISO_ITER_INIT, chain == 0 && a.M == 0);

  uint64_t hash = 0xCBF29CE484222325ull;
  int accepted = 0, lagCounter = 0, noS = 0;
  RoundOpen ro(a);
  for (int m = 0; m < a.M; m++) {
    const bool opens = ro.at(a, m);   // a round's first iteration: no proposal terms (miso.c:866)
    for (int k = 0; k < K; k++) hash = (hash ^ static_cast<uint32_t>(S.cnt[k])) * 0x100000001B3ull;
    if (trace) for (int k = lane; k < K; k += 64) trace[(static_cast<size_t>(m) * a.C + chain) * K + k] = S.cnt[k];
    propose(S.alpha, S.alphaN, S.psiN, static_cast<uint32_t>(m), accept_word);
    const double rp = PE ? (rbad ? miso_u2d(0x7FF8000000000000ull) : static_cast<double>(rfix) * (1.0 / MISO_SFIX_SCALE)) : 0.0;
    const double pp = big_joint<PE>(S, S.psiN, rp, c, K, lane);
    const double pc = big_joint<PE>(S, S.psi, rp, c, K, lane);
    const double ptoCS = big_proposal(S, S.psi, S.alphaN, c, K, lane);
    const double ctoPS = big_proposal(S, S.psiN, S.alpha, c, K, lane);
    const double acceptP = !opens ? miso_det_exp(pp + ptoCS - (pc + ctoPS)) : miso_det_exp(pp - pc);
    const bool acc = (acceptP >= 1) || (miso_u01(accept_word) < acceptP);
    double cJS = pc;
    if (acc) {
      for (int k = lane; k < K; k += 64) { S.psi[k] = S.psiN[k]; S.alpha[k] = S.alphaN[k]; }
      cJS = pp; accepted++;
    }
    bsync();
    if (m >= a.B) {  // miso.c:882-893
      if (lagCounter == a.lag - 1) {
        const size_t col = static_cast<size_t>(noS) + chain;
        for (int k = lane; k < K; k += 64) samples[col * K + k] = S.psi[k];
        if (lane == 0) loglik[col] = cJS;
        noS += a.C;
        lagCounter = 0;
      } else {
        lagCounter++;
      }
    }
    gibbs(static_cast<uint32_t>(m), chain == 0 && m == a.M - 1);
  }
  for (int k = 0; k < K; k++) hash = (hash ^ static_cast<uint32_t>(S.cnt[k])) * 0x100000001B3ull;
  if (trace) for (int k = lane; k < K; k += 64) trace[(static_cast<size_t>(a.M) * a.C + chain) * K + k] = S.cnt[k];
  if (lane == 0) {
    ChainStats *st = reinterpret_cast<ChainStats *>(a.out_pool + E.off_stats) + chain;
    st->counts_hash = hash;
    st->accepted = accepted;
    st->hw_id = 0;
  }
}

template __global__ void sampler_big<false>(const KernelArgs);
template __global__ void sampler_big<true>(const KernelArgs);

}  // namespace miso

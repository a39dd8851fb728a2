// kernels_flatl.inl -- sampler_flatl<KC>: sampler_flat (kernels_flat.inl) with the Metropolis-Hastings step taken OUT of the
// wavefronts' flat passes: ONE wavefront of the workgroup runs it for all the workgroup's chains, one chain per LANE.
//
// Why (round 3's profile, profiles/r02_flat_phase_valu.txt, gpurun_out/flat_phase_r3.txt): in sampler_flat every wavefront runs
// the scalar step (miso.c:449-552, 243-307, 97-163) for its own 7-8 chains as a dozen sparsely filled flat passes -- six
// transcendental passes, a leader lane's serial sums in between, an LDS round trip and a fence after each: 195 (K = 5) to
// 404 (K = 10) VALU wave-instructions per chain-iteration and 35-39 % of a wavefront-iteration's time, and it does not
// amortise over more chains per wavefront.  One chain per lane is the reference's arithmetic written out serially
// (sampler_lane_k's form): ~(5K + 3) transcendental calls per wavefront for up to 64 chains at once, no passes, no
// leader, no fences; the workgroup's other wavefronts wait at a barrier while the other workgroups of the CU run their
// read loops.  Which wavefront does it rotates with the workgroup's number (wavefront w sits on SIMD w % 4: always
// wavefront 0 would put every workgroup's scalar step on SIMD 0).
//
// Everything else is sampler_flat's: the slices (device.hpp FlatLayout; an odd number of 8-byte words apart, so that 64
// lanes reading the same entry of 64 slices never collide on a bank), thresholds per (chain, class), the read loops
// (flat_units / flat_units_desc), the resolve; a chain that owns a workgroup (FLAT_WIDE) has ONE slice here and all four
// wavefronts' lanes add into its D_k.  Same arithmetic, orders, tie rules and RNG addresses: bit-identical results.
#pragma once
#include "kernels_flat.inl"
#include "lane_mh.hpp"

#pragma clang fp contract(off)

namespace miso {

#ifndef MISO_FLATL_WGS_SMALL
#define MISO_FLATL_WGS_SMALL 3
#endif
template <int KC>
__global__ __launch_bounds__(256, KC <= 8 ? MISO_FLATL_WGS_SMALL : 2) void sampler_flatl(const KernelArgs a) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int NC = a.nc, ks = a.kstride, cs = a.cstride, trow = flat_trow(ks);
  const FlatLayout L = flat_layout(ks, cs);
  const long wave_id = static_cast<long>(blockIdx.x) * 4 + wid;
  const int wt_first = __builtin_amdgcn_readfirstlane(a.wave_tab[2 * wave_id]), wt_n = __builtin_amdgcn_readfirstlane(a.wave_tab[2 * wave_id + 1]);
  const bool wide = (wt_n & FLAT_WIDE) != 0;   // uniform over the workgroup (runtime.hip flat_waves)
  const long first_slot = wt_first;
  const int ncw = wt_n & 0xFF;                 // 0: padding wavefront (it still meets every barrier)
  // a workgroup-wide chain has one slice, at the start of the workgroup's LDS, used by all four wavefronts
  unsigned char *wbase = smem_flat + (wide ? 0 : static_cast<size_t>(wid) * NC * L.bytes);
  const bool writes = !wide || wid == 0;
  const uint32_t k0 = static_cast<uint32_t>(a.seed), k1 = static_cast<uint32_t>(a.seed >> 32);
  const int mh_wave = static_cast<int>(blockIdx.x & 3u);

#define FD(s, off) reinterpret_cast<double *>(wbase + (s) * L.bytes + (off))
#define FI(s, off) reinterpret_cast<int *>(wbase + (s) * L.bytes + (off))
#define FU(s, off) reinterpret_cast<uint32_t *>(wbase + (s) * L.bytes + (off))
  const int PR = ks;   // buffer 0 of psi / alpha / lp / tb / lr = the current state and its cached logs, buffer 1 (+ ks) = the proposal

  // ---- set-up: every chain's constants and class table into its slice (a workgroup-wide chain: wavefront 0) ----
  int Kw = 0;
  for (int s = 0; s < ncw; s++) {
    const long slot = first_slot + s;
    const int ev = a.slot_event[slot / a.C];
    const uint32_t chain = static_cast<uint32_t>(slot % a.C);
    const DevEvent E = a.events[ev];
    const int K = E.K;
    Kw = max(Kw, K);
    if (!writes) continue;
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
    for (int k = ks + lane; k <= trow; k += 64) FI(s, L.dl)[k] = 0;
    for (int i = lane; i < CLS_WORDS * (E.n_dcls + 1); i += 64) FU(s, L.ctab)[i] = gt[i];
    for (int i = lane; i < cs * trow; i += 64) FU(s, L.thr)[i] = 0u;
    if (lane == 0) {
      int *mi = FI(s, L.misc);
      mi[MI_K] = K; mi[MI_NDRAW] = E.n_draw; mi[MI_NCLS] = E.n_dcls; mi[MI_NUNITS] = E.n_units;
      mi[MI_EVID] = static_cast<int>(event_id); mi[MI_CHAIN] = static_cast<int>(chain);
      mi[MI_ACC] = 0; mi[MI_ACCW] = 0; mi[MI_EV] = ev; mi[MI_NEXT] = -1;
      mi[MI_DESC] = static_cast<int>(static_cast<uint32_t>(E.off_units >> 2));
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
  Kw = __builtin_amdgcn_readfirstlane(Kw);
  __syncthreads();
  // the wavefront's unit list: chain s owns units [ustart_s, ustart_s + n_units_s); MI_NEXT = the next chain with units
  // (wide: every wavefront would write the same words into the one slice: wavefront 0 does)
  int total_units = 0;
  {
    int nxt = -1;
    for (int s = ncw - 1; s >= 0; s--) {
      if (lane == 0 && writes) FI(s, L.misc)[MI_NEXT] = nxt;
      if (FI(s, L.misc)[MI_NUNITS] > 0) nxt = s;
    }
    for (int s = 0; s < ncw; s++) {
      if (lane == 0 && writes) FI(s, L.misc)[MI_USTART] = total_units;
      total_units += FI(s, L.misc)[MI_NUNITS];
    }
  }
  total_units = __builtin_amdgcn_readfirstlane(total_units);
  const int trips = (total_units + 63) / 64;
  const bool use_desc = a.flat_desc != 0;
  if (use_desc && total_units > 0 && lane == 0 && !wide) {   // lanes per chain for flat_units_desc (kernels_flat.inl)
    int sum = 0;
    for (int s = 0; s < ncw; s++) {
      const int nu = FI(s, L.misc)[MI_NUNITS];
      const int gs = nu > 0 ? max(1, static_cast<int>((64L * nu) / total_units)) : 0;
      FI(s, L.misc)[MI_LANES] = gs; sum += gs;
    }
    while (sum != 64) {
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
  int d_ms = 0, d_r = 0, d_g = 0;
  if (use_desc && total_units > 0 && !wide) {
    for (int s = 0; s < ncw; s++) {
      const int l0 = FI(s, L.misc)[MI_LANE0], gs = FI(s, L.misc)[MI_LANES];
      if (lane >= l0 && lane < l0 + gs) { d_ms = s; d_r = lane - l0; d_g = gs; }
    }
  }
  if (wide) { d_ms = 0; d_r = wid * 64 + lane; d_g = 256; }
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

  // ---- the Metropolis-Hastings wavefront: lane l owns chain (l / NC, l % NC) of the workgroup ----
  const bool mh_wv = wid == mh_wave;
  bool mh = false;
  unsigned char *mb = smem_flat;   // the lane's chain's slice
  {
    const int cw = lane / NC, csl = lane - cw * NC;
    if (mh_wv && cw < 4) {
      const int n_of = a.wave_tab[2 * (static_cast<long>(blockIdx.x) * 4 + cw) + 1];
      mh = wide ? lane == 0 : csl < (n_of & 0xFF);
      if (mh && !wide) mb = smem_flat + (static_cast<size_t>(cw) * NC + csl) * L.bytes;
    }
  }
  // the wavefront's largest isoform count (a scalar loop bound; lanes with fewer isoforms are switched off inside)
  int Kmh = 1;
  if (mh_wv) {
    int kk = mh ? reinterpret_cast<const int *>(mb + L.misc)[MI_K] : 1;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) kk = max(kk, __shfl_xor(kk, o));
    Kmh = __builtin_amdgcn_readfirstlane(kk);
  }
  // (lane_mh.hpp: the reference's step written out for one chain per lane; vectors AND state in the chain's slice)
  auto mh_ctx = [&]() {
    LaneMh c{};
    c.mb = mb;
    c.o = LaneMhOff{L.alpha, L.psi, L.lp, L.tb, L.lr, L.tc, L.cst, L.isc, L.hm1, L.bas, L.cnt, L.sx};
    const int *mi = reinterpret_cast<const int *>(mb + L.misc);
    c.PR = PR; c.lK = mh ? mi[MI_K] : 1; c.Kmh = Kmh;
    c.seed = a.seed; c.evid = static_cast<uint32_t>(mi[MI_EVID]); c.chain = static_cast<uint32_t>(mi[MI_CHAIN]);
    return c;
  };
  int lagCounter = 0, noS = 0;
#ifdef MISO_K2_PROFILE   // per-phase cycles of the Metropolis-Hastings wavefront (tools/phase_prof_flat.py): MH, thresholds, read loop + resolve, all
  uint64_t fp_mh = 0, fp_thr = 0, fp_loop = 0, fp_all = 0;
#endif

  auto count_of = [&](int s, int k) { return FI(s, L.bas)[k] + FI(s, L.cnt)[k]; };
  const float inv_k = 1.0f / static_cast<float>(ks), inv_cs = 1.0f / static_cast<float>(max(cs, 1));
#define FLAT_BEGIN(nper, inv)                                                   \
  for (int base_ = 0; base_ < ncw * (nper); base_ += 64) {                      \
    const int idx_ = base_ + lane;                                              \
    const bool on = idx_ < ncw * (nper);                                        \
    const int s = on ? static_cast<int>((static_cast<float>(idx_) + 0.5f) * (inv)) : 0; \
    const int j = on ? idx_ - s * (nper) : 0;
#define FLAT_END }

  // ---- per-read picks by direct evaluation of the reference's scan (miso.c:11-22, 69-80): chain 0's final assignment
  // (miso.c:943-946) and the fallback when a threshold does not fit 32 bits ----
  auto direct_chain = [&](int s, uint32_t iter, bool count, bool write) {
    const int *mi = FI(s, L.misc);
    const int K = mi[MI_K], n_draw = mi[MI_NDRAW];
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

  // ---- Gibbs step for every chain's current psi (miso.c:30-91); ends with a barrier: the counts are in the slices ----
  auto gibbs = [&](uint32_t iter) {
    // thresholds: one lane per (chain, class); a workgroup-wide chain's four wavefronts each compute the same rows
    // (so that each knows `slow` by itself) and write the same words
    FPROF_T(g0);
    bool slow = false;
    FLAT_BEGIN(cs, inv_cs)
      const int *mi = FI(s, L.misc);
      const int K = mi[MI_K], ncls = mi[MI_NCLS];
      const uint32_t m = FU(s, L.ctab)[CLS_WORDS * j];
      const double *psi = FD(s, L.psi);
      double ps[KC];
#pragma unroll
      for (int k = 0; k < KC; k++) ps[k] = (k < Kw) ? psi[k] : 0.0;
      const bool mine = on && j < ncls;
      double T = 0.0;   // total weight, ascending isoforms (miso.c:11-22); +0.0 for the others leaves the bits alone
#pragma unroll
      for (int k = 0; k < KC; k++) if (k < Kw) T = T + ((k < K && ((m >> k) & 1u)) ? ps[k] : 0.0);
      const double inv = 4294967296.0 / T;
      const bool tnormal = T >= 1e-280 && T <= 1e280;
      const bool le = __popc(m) != 2;
      const int kmax = 31 - __clz(static_cast<int>(m));
      uint32_t *th = FU(s, L.thr) + j * trow;
      double cum = 0.0;
      uint32_t run = 0u;
#pragma unroll
      for (int k = 0; k < KC - 1; k++) {
        if (k < Kw - 1) {
          const bool member = k < K && ((m >> k) & 1u);
          cum = cum + (member ? ps[k] : 0.0);
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
    FLAT_END
    if (wide) __syncthreads(); else fsync();
    FPROF_T(g1);
    FPROF_ADD(fp_thr, g0, g1);
    if (__any(slow)) {   // a non-final threshold of 2^32 cannot be held in 32 bits: direct path this time
      if (writes) {
        FLAT_BEGIN(ks, inv_k)
          if (on) FI(s, L.cnt)[j] = 0;
        FLAT_END
        fsync();
        for (int s = 0; s < ncw; s++) direct_chain(s, iter, true, false);
      }
      __syncthreads();
      return;
    }
    const int tww = Kw - 1;
    FlatUnitsArgs ua;
    ua.woff = wide ? 0 : wid * NC * L.bytes; ua.slice = L.bytes; ua.off_ctab = L.ctab; ua.off_thr = L.thr; ua.off_misc = L.misc; ua.off_dl = L.dl;
    ua.trow = trow; ua.trips = trips; ua.iter = iter; ua.k0 = k0; ua.k1 = k1;
#define MISO_FUNITS(TW) { if (use_desc || wide) flat_units_desc<TW>(ua, reinterpret_cast<const uint32_t *>(a.in_pool), d_ms, d_r, d_g); else flat_units<TW>(ua, s0, c0, i0, n_mine); }
    if (ncw > 0) {
      if constexpr (KC == 4) { if (tww <= 2) MISO_FUNITS(2) else MISO_FUNITS(3) }
      else if constexpr (KC == 8) { if (tww <= 4) MISO_FUNITS(4) else if (tww == 5) MISO_FUNITS(5) else if (tww == 6) MISO_FUNITS(6) else MISO_FUNITS(7) }
      else if constexpr (KC == 12) { if (tww <= 9) MISO_FUNITS(9) else MISO_FUNITS(11) }
      else if constexpr (KC == 16) { MISO_FUNITS(15) }
      else { if (tww <= 19) MISO_FUNITS(19) else if (tww <= 23) MISO_FUNITS(23) else MISO_FUNITS(31) }
    }
#undef MISO_FUNITS
    if (wide) __syncthreads(); else fsync();   // (wide: all four wavefronts' lanes have added into the one slice's D_k)
    if (writes) {
      // D_k (+ the reads of classes that end at or before k) -> picks per isoform
      FLAT_BEGIN(ks, inv_k)
        const int *mi = FI(s, L.misc);
        const int K = mi[MI_K], nd = mi[MI_NDRAW];
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
    }
    FPROF_T(g2);
    FPROF_ADD(fp_loop, g1, g2);
    __syncthreads();
  };

  // ---- initial state: miso.c:834 (alpha + sd z in place), cached logs, log-sum-exp, miso.c:841 ----
  if (mh) {
    LaneMh c = mh_ctx();
    const DevEvent LE_ = a.events[reinterpret_cast<const int *>(mb + L.misc)[MI_EV]];
    lane_mh_init(c, reinterpret_cast<const double *>(a.in_pool + LE_.off_consts) + 3 * c.lK);
  }
  __syncthreads();
  gibbs(MISO_ITER_INIT);

  const bool tracing = a.n_slots > 0 && a.events[a.slot_event[0]].off_trace != NO_TRACE;   // all events of a batch trace or none
  for (int m = 0; m < a.M; m++) {
    if (tracing && writes) {
      FLAT_BEGIN(ks, inv_k)
        const int *mi = FI(s, L.misc);
        const int K = mi[MI_K];
        if (on && j < K) {
          const uint64_t to = (static_cast<uint64_t>(static_cast<uint32_t>(mi[MI_TRACE_HI])) << 32) | static_cast<uint32_t>(mi[MI_TRACE_LO]);
          reinterpret_cast<int32_t *>(a.out_pool + to)[(static_cast<size_t>(m) * a.C + mi[MI_CHAIN]) * K + j] = count_of(s, j);
        }
      FLAT_END
    }
    const bool rec = m >= a.B && lagCounter == a.lag - 1;
    FPROF_T(m0);
    if (mh) {
      LaneMh c = mh_ctx();
      const double cJS = lane_mh_step<false>(c, m, 0.0);   // miso.c:449-552, 243-307, 869-880
      if (rec) {   // miso.c:882-893
        const DevEvent LE_ = a.events[reinterpret_cast<const int *>(mb + L.misc)[MI_EV]];
        double *l_samples = reinterpret_cast<double *>(a.out_pool + LE_.off_samples);
        const size_t col = static_cast<size_t>(noS) + c.chain;
        const LaneVec<double> psi = c.D(c.o.psi);
        for (int k = 0; k < c.lK; k++) l_samples[col * c.lK + k] = psi[k];
        reinterpret_cast<double *>(a.out_pool + LE_.off_loglik)[col] = cJS;
      }
    }
    if (m >= a.B) {
      if (rec) { noS += a.C; lagCounter = 0; } else lagCounter++;
    }
    FPROF_T(m1);
    FPROF_ADD(fp_mh, m0, m1);
    __syncthreads();
    gibbs(static_cast<uint32_t>(m));
    FPROF_T(m2);
    FPROF_ADD(fp_all, m0, m2);
  }
  if (mh) {
    LaneMh c = mh_ctx();
    lane_mh_begin(c);
    const LaneVec<int> bas = c.I(c.o.bas), cnt = c.I(c.o.cnt);
    for (int k = 0; k < c.lK; k++) c.hash = (c.hash ^ static_cast<uint32_t>(bas[k] + cnt[k])) * 0x100000001B3ull;
    lane_mh_end(c);
  }
  if (tracing && writes) {
    FLAT_BEGIN(ks, inv_k)
      const int *mi = FI(s, L.misc);
      const int K = mi[MI_K];
      if (on && j < K) {
        const uint64_t to = (static_cast<uint64_t>(static_cast<uint32_t>(mi[MI_TRACE_HI])) << 32) | static_cast<uint32_t>(mi[MI_TRACE_LO]);
        reinterpret_cast<int32_t *>(a.out_pool + to)[(static_cast<size_t>(a.M) * a.C + mi[MI_CHAIN]) * K + j] = count_of(s, j);
      }
    FLAT_END
  }
  // chain 0's final picks, read by read (miso.c:943-946): the last Gibbs step's draws once more
  for (int s = 0; s < ncw; s++)
    if (FI(s, L.misc)[MI_CHAIN] == 0 && writes)
      direct_chain(s, a.M > 0 ? static_cast<uint32_t>(a.M - 1) : MISO_ITER_INIT, false, true);
  if (mh) {
    LaneMh c = mh_ctx();
    lane_mh_begin(c);
    const DevEvent LE_ = a.events[reinterpret_cast<const int *>(mb + L.misc)[MI_EV]];
#ifdef MISO_K2_PROFILE
    if (c.chain == 0 && a.M > 8) {
      double *l_loglik = reinterpret_cast<double *>(a.out_pool + LE_.off_loglik);
      l_loglik[0] = static_cast<double>(fp_mh); l_loglik[1] = static_cast<double>(fp_thr); l_loglik[2] = static_cast<double>(fp_loop); l_loglik[3] = static_cast<double>(fp_all);
    }
#endif
    ChainStats *l_stats = reinterpret_cast<ChainStats *>(a.out_pool + LE_.off_stats) + c.chain;
    l_stats->counts_hash = c.hash; l_stats->accepted = c.accepted;
    l_stats->hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#undef FD
#undef FI
#undef FU
#undef FLAT_BEGIN
#undef FLAT_END
}

}  // namespace miso

// kernels_pel.inl -- sampler_pel<G, KC>: paired-end genes of three to twenty isoforms (BASELINE configs[3]; the reference's
// splicing_miso_paired, miso_paired.c:241-574), sampler_grp's normal size buckets re-divided:
//   * the Gibbs step (miso_paired.c:24-86) is sampler_grp's: G lanes per chain stride over the chain's dense records
//     (pe_dense, kernels_grp.inl) -- but the records are staged ONCE in the chain's LDS slice as far as the workgroup's LDS
//     allows (all of them up to ~6 isoforms at 16 lanes per chain, about half at ten) instead of being streamed from the
//     L2 / infinity cache every iteration (round 3: 3.5 - 4.3 TB per launch to re-read < 1 GB, 0.42 - 0.71 of the HBM peak);
//   * the Metropolis-Hastings step (miso.c:449-552 with miso_paired.c:88-174) is NOT run by every wavefront for its own
//     four chains as six sparsely filled transcendental passes (27 % of a wavefront-iteration at K = 5, 28 % at K = 10,
//     the same cost for two chains as for eight): ONE wavefront of the workgroup runs it for all the workgroup's chains,
//     one chain per lane (lane_mh.hpp), between two barriers; which wavefront rotates with the workgroup's number.
// Same arithmetic, orders, tie rules and RNG addresses as sampler_grp and the CPU checker's counter mode: bit-identical.
#pragma once
#include "kernels_grp.inl"
#include "lane_mh.hpp"
#include "pel_layout.hpp"

#pragma clang fp contract(off)

namespace miso {

template <int G, int KC>
__device__ __forceinline__ void pel_body(const KernelArgs &a, unsigned block_x) {
  constexpr int KLO = KC == 4 ? 3 : (KC == 8 ? 5 : (KC == 12 ? 9 : (KC == 16 ? 13 : 17)));   // the class holds K in [KLO, KC]
  constexpr int CPW = 64 / G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int il2 = pe_dense_il2(a.il);
  const int fp_bytes = (il2 * 8 + 15) & ~15;
  double *lds_fp = reinterpret_cast<double *>(smem);
  for (int i = threadIdx.x; i < il2; i += blockDim.x) lds_fp[i] = i < a.il ? a.frag_prob[i] : (i == a.il ? -0.0 : 1.0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane / G, sub = lane - grp * G;
  const long n_chains = static_cast<long>(a.n_slots) * a.C;
  const long first_chain = static_cast<long>(block_x) * 4 * CPW;          // of the workgroup
  long slot = first_chain + wave * CPW + grp;
  const bool live = slot < n_chains;
  if (!live) slot = n_chains - 1;       // shadow a real chain in this group's own slice, store nothing
  const int ks = a.kstride;
  const PelLayout L = pel_layout(ks, a.pel_ts, a.pel_rd);
  unsigned char *sb = smem + fp_bytes + static_cast<size_t>(wave * CPW + grp) * L.bytes;
#define SD(off) reinterpret_cast<double *>(sb + (off))
#define SI(off) reinterpret_cast<int *>(sb + (off))

  const int ev = a.slot_event[slot / a.C];
  const uint32_t chain = static_cast<uint32_t>(slot % a.C);
  const DevEvent E = a.events[ev];
  const int K = E.K;
  int nqw = (E.n_draw + 3) >> 2;    // wave-uniform loop bound of the read loop
  for (int off = 32; off >= 1; off >>= 1) nqw = max(nqw, __shfl_xor(nqw, off));
  nqw = __builtin_amdgcn_readfirstlane(nqw);
  if (!__all(K >= KLO && K <= (KC == 32 ? 20 : KC) && E.off_dense != NO_DENSE)) __builtin_trap();   // the host sends only such genes here
  if (static_cast<uint32_t>(reinterpret_cast<uintptr_t>(smem)) != 0u) __builtin_trap();             // pe_dense reads the LDS by byte address
  const uint32_t event_id = E.has_id ? E.explicit_id : a.first_event_id + static_cast<uint32_t>(ev);
  const double *consts = reinterpret_cast<const double *>(a.in_pool + E.off_consts);
  const int *base = reinterpret_cast<const int *>(a.in_pool + E.off_base);
  const uint32_t *dq = reinterpret_cast<const uint32_t *>(a.in_pool + E.off_dense);
  const int32_t *sfixd_glob = reinterpret_cast<const int32_t *>(a.in_pool + E.off_sfixd);
  const int n_draw = E.n_draw, n_quads = (n_draw + 3) >> 2;
  const int ND = K + 1;             // dwords per quad of dense records
  const int nql = min(n_quads + 1, a.pel_rd / ND);   // quads (incl. the padding quad after the last) staged in LDS
  // ---- set-up: the chain's G lanes fill its slice ----
  for (int k = sub; k < ks; k += G) {
    const bool in = k < K;
    SD(L.cst)[k] = in ? consts[k] : 0.0; SD(L.hm1)[k] = in ? consts[2 * K + k] : 0.0;
    // miso.c:330-447 START_AUTO: K != 2 -> alpha = 1/(K-1); START_UNIFORM -> 0
    SD(L.alpha)[k] = (a.start == MISO_START_AUTO && K != 2 && k < K - 1) ? 1.0 / (K - 1) : 0.0;
    SD(L.alpha)[ks + k] = 0.0; SD(L.psi)[k] = 0.0; SD(L.psi)[ks + k] = 0.0;
    SD(L.lp)[k] = 0.0; SD(L.lp)[ks + k] = 0.0; SD(L.tb)[k] = 0.0; SD(L.tb)[ks + k] = 0.0;
    SD(L.lr)[k] = 0.0; SD(L.lr)[ks + k] = 0.0; SD(L.tc)[k] = 0.0;
    SI(L.cnt)[k] = 0; SI(L.bas)[k] = in ? base[k] : 0; SI(L.dl)[k] = 0;
  }
  if (sub == 0) {
    int *mi = SI(L.misc);
    mi[PM_K] = K; mi[PM_NDRAW] = n_draw; mi[PM_EV] = ev; mi[PM_CHAIN] = static_cast<int>(chain);
    mi[PM_EVID] = static_cast<int>(event_id); mi[PM_RBAD] = 0; mi[PM_NQL] = nql;
    *reinterpret_cast<int64_t *>(sb + L.rfix) = 0;
  }
  if (a.pel_ts > 0) { int32_t *st = reinterpret_cast<int32_t *>(sb + L.stab); for (int i = sub; i < K * il2; i += G) st[i] = sfixd_glob[i]; }
  { uint32_t *rc = reinterpret_cast<uint32_t *>(sb + L.rec); for (int i = sub; i < nql * ND; i += G) rc[i] = dq[i]; }
  __syncthreads();
  const uint32_t stab_lds = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(sb + L.stab));
  const uint32_t *recl = reinterpret_cast<const uint32_t *>(sb + L.rec);   // (a generic pointer into the LDS aperture)
  const bool dense_nobad = __all(E.dense_nobad != 0);
  uint8_t *drawass = a.out_pool + E.off_drawass;
  int32_t *trace = (E.off_trace == NO_TRACE) ? nullptr : reinterpret_cast<int32_t *>(a.out_pool + E.off_trace);
  const GibbsRng rng = gibbs_rng_init(a.seed, event_id, chain);

  // ---- the Metropolis-Hastings wavefront: lane l owns the workgroup's chain l (slice l) ----
  const bool mh_wv = wave == static_cast<int>(block_x & 3u);
  const bool mh = mh_wv && lane < 4 * CPW && first_chain + lane < n_chains;
  static_assert(PEL_ST >= LANE_MH_ST, "state block");
  unsigned char *mb = smem + fp_bytes + static_cast<size_t>(mh ? lane : 0) * L.bytes;
  int Kmh = 1;
  {
    int kk = mh ? reinterpret_cast<const int *>(mb + L.misc)[PM_K] : 1;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) kk = max(kk, __shfl_xor(kk, o));
    Kmh = __builtin_amdgcn_readfirstlane(kk);
  }
  // (lane_mh.hpp: vectors AND state in the chain's slice; rebuilt from the slice every time, nothing kept in registers
  // across the read loops)
  auto mh_ctx = [&]() {
    LaneMh c{};
    c.mb = mb;
    c.o = LaneMhOff{L.alpha, L.psi, L.lp, L.tb, L.lr, L.tc, L.cst, L.cst, L.hm1, L.bas, L.cnt, L.st};
    const int *mi = reinterpret_cast<const int *>(mb + L.misc);
    c.PR = ks; c.lK = mh ? mi[PM_K] : 1; c.Kmh = Kmh;
    c.seed = a.seed; c.evid = static_cast<uint32_t>(mi[PM_EVID]); c.chain = static_cast<uint32_t>(mi[PM_CHAIN]);
    return c;
  };

  // ---- Gibbs step for the chain's current psi (miso_paired.c:24-86); the caller's barrier follows ----
  auto gibbs = [&](uint32_t iter, bool write_ass) __attribute__((always_inline)) {
    int *dl = SI(L.dl);
    for (int k = sub; k < K; k += G) dl[k] = 0;
    wave_sync();
    int64_t acc = 0; int bad = 0;
    const uint32_t n0r0 = rng.p1hi ^ iter ^ rng.k0;
    const double *psi = SD(L.psi);
#define MISO_PED2(KK, LDS)                                                                            \
  {                                                                                                   \
    if (__any(write_ass)) pe_dense<KK, G, true, true, LDS, true>(dq, psi, lds_fp, stab_lds, sfixd_glob, il2, dl, drawass, write_ass, nqw, n_quads, n_draw, sub, rng, n0r0, a.pe_force_exact != 0, acc, bad, 0, recl, nql); \
    else if (dense_nobad) pe_dense<KK, G, false, false, LDS, true>(dq, psi, lds_fp, stab_lds, sfixd_glob, il2, dl, drawass, write_ass, nqw, n_quads, n_draw, sub, rng, n0r0, a.pe_force_exact != 0, acc, bad, 0, recl, nql); \
    else pe_dense<KK, G, false, true, LDS, true>(dq, psi, lds_fp, stab_lds, sfixd_glob, il2, dl, drawass, write_ass, nqw, n_quads, n_draw, sub, rng, n0r0, a.pe_force_exact != 0, acc, bad, 0, recl, nql); \
  }
#define MISO_PED(KK) { if (a.pel_ts > 0) MISO_PED2(KK, true) else MISO_PED2(KK, false) }
    if constexpr (KC == 4) { if (K == 3) MISO_PED(3) else MISO_PED(4) }
    else if constexpr (KC == 8) { if (K == 5) MISO_PED(5) else if (K == 6) MISO_PED(6) else if (K == 7) MISO_PED(7) else MISO_PED(8) }
    else if constexpr (KC == 12) { if (K == 9) MISO_PED(9) else if (K == 10) MISO_PED(10) else if (K == 11) MISO_PED(11) else MISO_PED(12) }
    else if constexpr (KC == 16) { if (K == 13) MISO_PED(13) else if (K == 14) MISO_PED(14) else if (K == 15) MISO_PED(15) else MISO_PED(16) }
    else { if (K == 17) MISO_PED(17) else if (K == 18) MISO_PED(18) else if (K == 19) MISO_PED(19) else MISO_PED(20) }
#undef MISO_PED2
#undef MISO_PED
    wave_sync();
#pragma unroll
    for (int off = G >> 1; off >= 1; off >>= 1) { acc += __shfl_xor(acc, off); bad |= __shfl_xor(bad, off); }
    // reads that passed over k - 1 but not k picked k
    for (int k = sub; k < K; k += G) SI(L.cnt)[k] = (k > 0 ? dl[k - 1] : n_draw) - (k < K - 1 ? dl[k] : 0);
    if (sub == 0) {
      *reinterpret_cast<int64_t *>(sb + L.rfix) = E.base_sfix + acc;
      SI(L.misc)[PM_RBAD] = bad | E.base_bad;
    }
  };

  // ---- initial state: miso.c:330-447, 834, 841 ----
  if (mh) {
    LaneMh c = mh_ctx();
    const DevEvent LE = a.events[reinterpret_cast<const int *>(mb + L.misc)[PM_EV]];
    lane_mh_init(c, reinterpret_cast<const double *>(a.in_pool + LE.off_consts) + 3 * c.lK);
  }
  __syncthreads();
  gibbs(MISO_ITER_INIT, live && chain == 0 && a.M == 0);
  __syncthreads();

  int lagCounter = 0, noS = 0;
  for (int m = 0; m < a.M; m++) {
    if (trace && live)
      for (int k = sub; k < K; k += G) trace[(static_cast<size_t>(m) * a.C + chain) * K + k] = SI(L.bas)[k] + SI(L.cnt)[k];
    const bool rec = m >= a.B && lagCounter == a.lag - 1;
    if (mh) {
      LaneMh c = mh_ctx();
      const int rbad = reinterpret_cast<const int *>(mb + L.misc)[PM_RBAD];
      const int64_t rfix = *reinterpret_cast<const int64_t *>(mb + L.rfix);
      // miso_paired.c:157-163: the picks' fragment scores, summed in 2^-26 fixed point; one non-finite entry makes the score NaN
      const double rp = rbad ? miso_u2d(0x7FF8000000000000ull) : static_cast<double>(rfix) * (1.0 / MISO_SFIX_SCALE);
      const double cJS = lane_mh_step<true>(c, m, rp);
      if (rec) {   // miso.c:882-893
        const DevEvent LE = a.events[reinterpret_cast<const int *>(mb + L.misc)[PM_EV]];
        double *l_samples = reinterpret_cast<double *>(a.out_pool + LE.off_samples);
        const size_t col = static_cast<size_t>(noS) + c.chain;
        const LaneVec<double> psi = c.D(c.o.psi);
        for (int k = 0; k < c.lK; k++) l_samples[col * c.lK + k] = psi[k];
        reinterpret_cast<double *>(a.out_pool + LE.off_loglik)[col] = cJS;
      }
    }
    if (m >= a.B) {
      if (rec) { noS += a.C; lagCounter = 0; } else lagCounter++;
    }
    __syncthreads();
    gibbs(static_cast<uint32_t>(m), live && chain == 0 && m == a.M - 1);
    __syncthreads();
  }
  if (trace && live)
    for (int k = sub; k < K; k += G) trace[(static_cast<size_t>(a.M) * a.C + chain) * K + k] = SI(L.bas)[k] + SI(L.cnt)[k];
  if (mh) {
    LaneMh c = mh_ctx();
    lane_mh_begin(c);
    const LaneVec<int> bas = c.I(c.o.bas), cnt = c.I(c.o.cnt);
    for (int k = 0; k < c.lK; k++) c.hash = (c.hash ^ static_cast<uint32_t>(bas[k] + cnt[k])) * 0x100000001B3ull;
    const DevEvent LE = a.events[reinterpret_cast<const int *>(mb + L.misc)[PM_EV]];
    ChainStats *l_stats = reinterpret_cast<ChainStats *>(a.out_pool + LE.off_stats) + c.chain;
    l_stats->counts_hash = c.hash; l_stats->accepted = c.accepted; l_stats->hw_id = 0;
  }
#undef SD
#undef SI
}

// two workgroups per CU (256 registers): pe_dense wants the registers more than the occupancy (kernels_grp.inl)
template <int G, int KC>
__global__ __launch_bounds__(256, 2) void sampler_pel(const KernelArgs a) {
  pel_body<G, KC>(a, blockIdx.x);
}

}  // namespace miso

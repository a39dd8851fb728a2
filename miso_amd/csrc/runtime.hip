// runtime.hip -- batch object: host packing -> HBM pools -> kernel launches -> results.
//
// One batch = the events one GPU samples in one go (the unit the reference hands to one worker
// process, misopy/miso.py:165-187).  All events of a batch share the sampler parameters
// (run_miso.py:68-73 passes the same settings for every event of a run).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

#include "batch.hpp"
#include "coop.hpp"
#include "miso_binomial.h"

namespace miso {

extern int g_hw_queues_in_effect;   // capi.hip: GPU_MAX_HW_QUEUES as the runtime read (or will read) it, 0 = unknown

template <bool PE> __global__ void sampler_wave(const KernelArgs a);
template <int G, int MODE, int WPB> __global__ void sampler_k2(const KernelArgs a);
template <int GA, int GB> __global__ void sampler_k2_mix(const KernelArgs a);
template <int MODE, int WPB, bool NARROW = false> __global__ void sampler_k2_multi(const KernelArgs a);
template <bool PE> __global__ void sampler_big(const KernelArgs a);   // kernels_big.hip: 65 ... MISO_MAX_ISOFORMS isoforms
__global__ void sampler_lane(const KernelArgs a);   // kernels_lane.hip: collapsed Gibbs step, one chain per lane
__global__ void sampler_lane_ilp(const KernelArgs a);   // ... the form for at most two wavefronts per SIMD
template <int G> __global__ void sampler_k2c(const KernelArgs a);   // ... G lanes per chain (k2_body COLLAPSED)
__global__ void sampler_lane_k(const KernelArgs a); // ... three and more isoforms (vectors in LDS)
__global__ void sampler_marginal(const KernelArgs a); // kernels_marginal.hip: algorithm = MARGINAL, one chain per lane
constexpr int LANEK_VECTORS = 9;
inline size_t lanek_lds_bytes(int ks) { return static_cast<size_t>(LANEK_VECTORS) * ks * 64 * 8 + static_cast<size_t>(ks) * 64 * 4; }
__global__ void compare_kernel(const DevEvent *, const unsigned char *, const DevEvent *, const unsigned char *, int, int,
                               double, const uint64_t *, double *);
__global__ void match_kernel(const MatchEvent *, const int2 *, const int *, const int *, const int *, const int *,
                             const int *, const int *, const int *, int, int, int, int, int, uint32_t *, uint16_t *);
__global__ void summarize_kernel(const DevEvent *, const unsigned char *, int, int, int, int, const uint64_t *, double *, int);
template <int G, bool PE, int KC, bool WIDE = false> __global__ void sampler_grp(const KernelArgs a);
template <int KC> __global__ void sampler_grp_multi(const KernelArgs a);
__global__ void sampler_grp_all(const KernelArgs a);   // kernels_grp_all.hip
template <int KC, int KS, bool UNI> __global__ void sampler_flat(const KernelArgs a);   // KS: the launch's largest isoform count at compile time (0: at run time)
__global__ void selftest_detmath_kernel(const double *, int, double *, double *, double *, double *);
__global__ void selftest_philox_kernel(const uint32_t *, int, uint32_t *);

#define HIP_OK(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      MISO_FAIL(MISO_ENODEVICE, std::string(#call) + ": " + hipGetErrorString(e_));        \
  } while (0)

// ---- the shader clock a launch ran at (miso_batch_set_clock_probe; bench.py's roofline prices its VALU peak with it) ----
// ONE wavefront beside the sampler kernels, on a stream of its own: it sleeps, looks at a flag the batch's stream sets
// behind its last kernel, and leaves with the two counters' values at both ends -- s_memtime counts shader cycles,
// wall_clock64() the constant reference clock (hipDeviceAttributeWallClockRate: 100 MHz here) -- so cycles / time over exactly the launch.  No VALU work, no LDS: it
// takes one wave slot of one SIMD.  The flag carries the launch's number (nothing to reset between launches).  It gives up after `max_ticks` of the reference clock (a flag that never comes must
// not hold the device).  out: {t0 real, t0 cycles, t1 real, t1 cycles, 1 = gave up}.
__global__ void clock_probe_kernel(const uint32_t *flag, uint32_t want, unsigned long long *out, unsigned long long max_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long r0 = static_cast<unsigned long long>(wall_clock64()), c0 = __builtin_readcyclecounter();
  unsigned long long r1 = r0;
  uint32_t f = 0;
  do {
    __builtin_amdgcn_s_sleep(127);
    f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r1 = static_cast<unsigned long long>(wall_clock64());
  } while (f != want && r1 - r0 < max_ticks);
  const unsigned long long c1 = __builtin_readcyclecounter();
  r1 = static_cast<unsigned long long>(wall_clock64());
  out[0] = r0; out[1] = c0; out[2] = r1; out[3] = c1; out[4] = f != want;
}
__global__ void clock_probe_stop(uint32_t *flag, uint32_t value) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void set_device(int d) { HIP_OK(hipSetDevice(d)); }

static inline uint64_t align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

// Lanes per chain for sampler_k2.  Measured on MI355X (profiles/r01_k2_phase_cycles.txt): the
// kernel is VALU-issue bound at two resident wavefronts per SIMD; the per-iteration scalar MH
// step costs a wavefront ~9k cycles whatever G is, so fewer lanes per chain (more chains sharing
// it) is better -- until the wavefronts no longer fit the device's resident slots and a second,
// mostly empty round starts.  Hence: the smallest supported G whose wavefront count still fills
// the slots once, i.e. the largest G with ceil(chains / (64 / G)) <= slots; G = 1 for batches that
// overflow anyway; never more lanes than a chain has pairs of draw quads to stride over.
static inline int flat_wgs_for(int kc) { return kc <= 4 ? 3 : (kc == 12 ? 3 : 2); }

int choose_lanes_per_chain(long chains, int max_quads, int wave_slots, int max_cpw) {
  static const int kG[] = {64, 32, 21, 16, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1};
  const int cap = std::max(1, max_quads / 2);
  const bool pow2_only = max_cpw < 64;   // paired-end reduces three values per step: measured, the
  int last = 64;                         // shuffle-loop reduction of odd group sizes does not pay
  for (int G : kG) {
    const int cpw = 64 / G;
    if (cpw > max_cpw) break;            // paired-end: per-chain LDS tables bound the chains per wave
    if (pow2_only && (G & (G - 1))) continue;
    last = G;
    if (G > cap) continue;
    if ((chains + cpw - 1) / cpw <= wave_slots) return G;
  }
  return last;
}

}  // namespace miso

using namespace miso;

bool miso_batch::coop_enabled() const { return !no_coop && std::getenv("MISO_NO_COOP") == nullptr; }

void miso_batch::release() {
  if (device >= 0) (void) hipSetDevice(device);
  if (d_events) (void) hipFree(d_events);
  if (d_in) (void) hipFree(d_in);
  if (d_out) (void) hipFree(d_out);
  if (d_fp) (void) hipFree(d_fp);
  if (d_slots) (void) hipFree(d_slots);
  if (d_logfact) (void) hipFree(d_logfact);
  if (d_k2_pair_tab) (void) hipFree(d_k2_pair_tab);
  d_slots = nullptr;
  // (the next upload may be to another device: nothing may outlive its allocation as a stale pointer or a cached plan)
  d_logfact = nullptr; logfact_n = 0;
  d_k2_pair_tab = nullptr; k2_pair_tab.clear(); k2_pair_grid = k2_pair_wide_blocks = 0;
  k2_plan_key = k2w_plan_key = -1;
  if (probe_stream) { (void) hipStreamSynchronize(probe_stream); (void) hipStreamDestroy(probe_stream); probe_stream = nullptr; }
  if (d_probe) { (void) hipFree(d_probe); d_probe = nullptr; }
  if (d_round_tab) { (void) hipFree(d_round_tab); d_round_tab = nullptr; }
  if (d_grp_segs) { (void) hipFree(d_grp_segs); d_grp_segs = nullptr; }
  probe_armed = false;
  if (ev0) (void) hipEventDestroy(ev0);
  if (ev1) (void) hipEventDestroy(ev1);
  for (GenRun &run : gen_runs) {
    if (run.d_wave_tab) (void) hipFree(run.d_wave_tab);
    if (run.d_coop_tab) (void) hipFree(run.d_coop_tab);
    if (run.d_coop_mem) (void) hipFree(run.d_coop_mem);
    run.d_wave_tab = nullptr; run.d_coop_tab = nullptr; run.d_coop_mem = nullptr; run.wave_key = -1;
  }
  for (K2Coop *cc : {&k2_coop_se, &k2w_coop}) {
    if (cc->d_tab) (void) hipFree(cc->d_tab);
    if (cc->d_mem) (void) hipFree(cc->d_mem);
    *cc = K2Coop{};
  }
  for (hipStream_t st : aux_streams) (void) hipStreamDestroy(st);
  for (hipEvent_t e : aux_done) (void) hipEventDestroy(e);
  aux_streams.clear(); aux_done.clear();
  if (stream) (void) hipStreamDestroy(stream);
  d_events = nullptr; d_in = d_out = nullptr; d_fp = nullptr; ev0 = ev1 = nullptr; stream = nullptr;
  uploaded = launched = downloaded = false;
}

// Row f1: the compatibility of every pending event's reads with its isoforms, in one launch.
// Inputs: the genes' exon tables and the host-parsed CIGAR blocks; outputs: u32 masks (single-end)
// or u16 fragment indices (paired-end), from which the events are packed exactly as the host path
// packs them (the same pack_event, fed with the match matrix those outputs stand for).
void miso_batch::resolve_pending() {
  if (pending.empty()) return;
  const bool timing = std::getenv("MISO_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count(); };
  const auto T0 = now();
  const int n = static_cast<int>(pending.size());
  const bool pe = p.paired != 0;
  const int mates = pe ? 2 : 1;
  const int ov = p.overHang == 0 ? 1 : p.overHang;
  std::vector<MatchEvent> mev(n);
  std::vector<int2> blocks;
  std::vector<int> h_exidx, h_exstart, h_exend, h_pos, h_opidx, h_ops, h_len;
  // offsets first, then the copies in parallel
  std::vector<size_t> o_xi(n + 1, 0), o_ex(n + 1, 0), o_rd(n + 1, 0), o_op(n + 1, 0), o_out(n + 1, 0);
  std::vector<size_t> frag_off(n, 0);
  size_t out_frags = 0;
  for (int i = 0; i < n; i++) {
    const Pending &q = pending[i];
    const int N = events[q.event].N;
    const size_t K = static_cast<size_t>(q.gene.K);
    o_xi[i + 1] = o_xi[i] + q.gene.exidx.size();
    o_ex[i + 1] = o_ex[i] + q.gene.exstart.size();
    o_rd[i + 1] = o_rd[i] + static_cast<size_t>(mates) * N;
    o_op[i + 1] = o_op[i] + q.ct.ops.size();
    // frags[(out_off + r) * K + k]: out_off in units of the event's own K, so round up
    const size_t slot = pe ? (out_frags + K - 1) / K : o_out[i];
    frag_off[i] = slot * K;
    if (pe) out_frags = frag_off[i] + static_cast<size_t>(N) * K;
    o_out[i + 1] = o_out[i] + static_cast<size_t>(N) * (pe ? 1 : (K + 31) / 32);   // single-end masks: (K + 31) / 32 words per read
    MatchEvent &m = mev[i];
    m.K = q.gene.K; m.n_reads = N;
    m.exidx_off = static_cast<int32_t>(o_xi[i]); m.ex_off = static_cast<int32_t>(o_ex[i]);
    m.read_off = static_cast<int32_t>(o_rd[i]); m.out_off = static_cast<int32_t>(slot);
    for (int r0 = 0; r0 < N; r0 += 256) blocks.push_back(make_int2(i, r0));
  }
  if (o_op[n] > 0x7FFFFFF0u || o_rd[n] > 0x7FFFFFF0u || out_frags > 0x7FFFFFF0u)
    MISO_FAIL(MISO_EINVAL, "Batch too large for one match launch");
  const size_t out_slots = o_out[n];
  h_exidx.resize(o_xi[n]); h_exstart.resize(o_ex[n]); h_exend.resize(o_ex[n]);
  h_pos.resize(o_rd[n]); h_len.resize(o_rd[n]); h_opidx.resize(o_rd[n] + 1); h_ops.resize(o_op[n]);
  const int nthreads = std::max(1, std::min<int>(16, static_cast<int>(std::thread::hardware_concurrency())));
  auto parallel = [&](auto &&fn) {
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++) pool.emplace_back([&fn, t] { fn(t); });
    fn(0);
    for (auto &th : pool) th.join();
  };
  parallel([&](int t) {
    for (int i = t; i < n; i += nthreads) {
      const Pending &q = pending[i];
      std::copy(q.gene.exidx.begin(), q.gene.exidx.end(), h_exidx.begin() + o_xi[i]);
      std::copy(q.gene.exstart.begin(), q.gene.exstart.end(), h_exstart.begin() + o_ex[i]);
      std::copy(q.gene.exend.begin(), q.gene.exend.end(), h_exend.begin() + o_ex[i]);
      std::copy(q.pos.begin(), q.pos.end(), h_pos.begin() + o_rd[i]);
      std::copy(q.ct.len.begin(), q.ct.len.end(), h_len.begin() + o_rd[i]);
      std::copy(q.ct.ops.begin(), q.ct.ops.end(), h_ops.begin() + o_op[i]);
      const int base = static_cast<int>(o_op[i]);
      const size_t nr = o_rd[i + 1] - o_rd[i];
      for (size_t r = 0; r < nr; r++) h_opidx[o_rd[i] + r] = base + q.ct.idx[r];
    }
  });
  h_opidx[o_rd[n]] = static_cast<int>(o_op[n]);
  const auto T1 = now();
  std::vector<uint32_t> h_masks(pe ? 0 : out_slots);
  std::vector<uint16_t> h_frags(pe ? out_frags : 0);
  if (!blocks.empty()) {
    auto up = [&](const void *src, size_t bytes) {
      void *d = nullptr;
      HIP_OK(hipMalloc(&d, std::max<size_t>(bytes, 16)));
      if (bytes) HIP_OK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
      return d;
    };
    void *d_ev = up(mev.data(), mev.size() * sizeof(MatchEvent));
    void *d_bl = up(blocks.data(), blocks.size() * sizeof(int2));
    void *d_xi = up(h_exidx.data(), h_exidx.size() * 4), *d_xs = up(h_exstart.data(), h_exstart.size() * 4);
    void *d_xe = up(h_exend.data(), h_exend.size() * 4), *d_po = up(h_pos.data(), h_pos.size() * 4);
    void *d_oi = up(h_opidx.data(), h_opidx.size() * 4), *d_op = up(h_ops.data(), h_ops.size() * 4);
    void *d_le = up(h_len.data(), h_len.size() * 4);
    void *d_out = nullptr;
    const size_t out_bytes_m = pe ? h_frags.size() * 2 : h_masks.size() * 4;
    HIP_OK(hipMalloc(&d_out, std::max<size_t>(out_bytes_m, 16)));
    hipEvent_t t0, t1;
    HIP_OK(hipEventCreate(&t0)); HIP_OK(hipEventCreate(&t1));
    HIP_OK(hipEventRecord(t0, nullptr));
    hipLaunchKernelGGL(match_kernel, dim3(static_cast<unsigned>(blocks.size())), dim3(256), 0, nullptr,
                       static_cast<const MatchEvent *>(d_ev), static_cast<const int2 *>(d_bl),
                       static_cast<const int *>(d_xi), static_cast<const int *>(d_xs),
                       static_cast<const int *>(d_xe), static_cast<const int *>(d_po),
                       static_cast<const int *>(d_oi), static_cast<const int *>(d_op),
                       static_cast<const int *>(d_le), p.readLength, ov, pe ? 1 : 0, fd.start,
                       static_cast<int>(fd.prob.size()), static_cast<uint32_t *>(d_out),
                       static_cast<uint16_t *>(d_out));
    HIP_OK(hipGetLastError());
    HIP_OK(hipEventRecord(t1, nullptr));
    HIP_OK(hipMemcpy(pe ? static_cast<void *>(h_frags.data()) : static_cast<void *>(h_masks.data()), d_out,
                     out_bytes_m, hipMemcpyDeviceToHost));
    HIP_OK(hipEventElapsedTime(&match_ms, t0, t1));
    (void) hipEventDestroy(t0); (void) hipEventDestroy(t1);
    for (void *d : {d_ev, d_bl, d_xi, d_xs, d_xe, d_po, d_oi, d_op, d_le, d_out}) (void) hipFree(d);
  }
  const auto T2 = now();
  if (p.want_counts_trace) { kept_masks.resize(events.size()); kept_frags.resize(events.size()); }
  // pack on the host cores (a few, the cgroup may allow far fewer than the machine has)
  std::vector<std::string> errors(nthreads);
  std::vector<int> codes(nthreads, 0);
  parallel([&](int t) {
    try {
      for (int i = t; i < n; i += nthreads) {
        const Pending &q = pending[i];
        const int K = q.gene.K, N = mev[i].n_reads;
        const uint32_t *mk32 = pe ? nullptr : h_masks.data() + mev[i].out_off;
        const uint16_t *f = pe ? h_frags.data() + frag_off[i] : nullptr;
        const int W = (K + 63) / 64, W32 = (K + 31) / 32;   // mask words per read: 64-bit on the host, 32-bit from the kernel
        std::vector<uint64_t> pm(static_cast<size_t>(std::max(N, 1)) * W, 0u);
        if (pe) {   // a pair's mask = the isoforms with a fragment length inside the distribution
          for (int r = 0; r < N; r++)
            for (int k = 0; k < K; k++) if (f[static_cast<size_t>(r) * K + k] != FRAG_NONE) pm[static_cast<size_t>(r) * W + (k >> 6)] |= 1ull << (k & 63);
        } else {
          for (int r = 0; r < N; r++)
            for (int w = 0; w < W32; w++) pm[static_cast<size_t>(r) * W + (w >> 1)] |= static_cast<uint64_t>(mk32[static_cast<size_t>(W32) * r + w]) << (32 * (w & 1));
        }
        const uint64_t *mk = pm.data();
        if (p.want_counts_trace) {
          if (pe) kept_frags[q.event].assign(f, f + static_cast<size_t>(N) * K);
          else kept_masks[q.event].assign(mk, mk + static_cast<size_t>(N) * W);
        }
        events[q.event] = pack_event_masks(p, pe ? &fd : nullptr, K, N, mk, f, nullptr, q.gene.isolen.data(),
                                           q.gene.noexons.data(), q.hyper.empty() ? nullptr : q.hyper.data(), W);
        attach_gene_classes(events[q.event], p, q.gene);   // algorithm = CLASSES only
      }
    } catch (const Error &e) { codes[t] = e.code; errors[t] = e.text;
    } catch (const std::bad_alloc &) { codes[t] = MISO_ENOMEM; errors[t] = "Error at runtime.hip:0: allocation failed, Out of memory";
    } catch (const std::exception &e) {   // nothing may leave a worker thread (std::terminate)
      codes[t] = MISO_EINTERNAL; errors[t] = std::string("Error at runtime.hip:0: ") + e.what() + ", Internal error";
    } catch (...) { codes[t] = MISO_EINTERNAL; errors[t] = "Error at runtime.hip:0: unknown exception, Internal error"; }
  });
  for (int t = 0; t < nthreads; t++) if (codes[t]) throw Error(codes[t], errors[t], Error::Formatted{});
  pending.clear();
  if (timing)
    std::fprintf(stderr, "[miso] resolve: flatten %.3f s, device (alloc+copies+kernel %.3f ms) %.3f s, pack (%d threads) %.3f s\n",
                 secs(T0, T1), match_ms, secs(T1, T2), nthreads, secs(T2, now()));
}

void miso_batch::upload(int dev) {
  if (uploaded && dev == device) return;
  if (uploaded) release();
  if (device_count() <= 0) MISO_FAIL(MISO_ENODEVICE, "no HIP device: the sampler has no CPU path");
  device = dev;
  HIP_OK(hipSetDevice(dev));
  resolve_pending();
  {   // the reference's proposal density overflows beyond about 85 isoforms (miso.c:97-122: linear domain):
      // such a chain accepts its first proposal and nothing after -- bit-equal to the reference, not a posterior
    int wide = 0, kmax = 0;
    for (const PackedEvent &e : events) if (e.K > 80) { wide++; kmax = std::max(kmax, e.K); }
    if (wide > 0 && std::getenv("MISO_QUIET") == nullptr)
      std::fprintf(stderr, "[miso] warning: %d gene(s) of more than 80 isoforms (up to %d): as in the reference, the proposal density "
                           "(miso.c:97-122, linear domain) overflows beyond about 85 isoforms and the chains stop accepting after "
                           "their first iteration; the samples equal the reference's and are not a posterior\n", wide, kmax);
  }
  const int n = static_cast<int>(events.size());
  const int C = p.noChains, M = p.noIterations, Sn = S();
  h_events.assign(n, DevEvent{});
  uint64_t in_off = 0, out_off = 0;
  for (int i = 0; i < n; i++) {
    const PackedEvent &e = events[i];
    DevEvent &d = h_events[i];
    d.K = e.K; d.n_draw = e.n_draw; d.n_reads = e.N; d.base_bad = e.base_bad;
    d.base_sfix = e.base_sfix;
    if (i < static_cast<int>(event_ids.size()) && event_ids[i] >= 0) {
      d.has_id = 1; d.explicit_id = static_cast<uint32_t>(event_ids[i]);
    }
    d.off_consts = in_off; in_off = align_up(in_off + e.consts.size() * 8, 16);
    d.off_base = in_off; in_off = align_up(in_off + e.base_count.size() * 4, 16);
    d.off_draw = in_off;   // (single-end: the masks' low words, whole quads; from 33 isoforms on the high words behind them)
    in_off = align_up(in_off + (e.paired ? e.draw_frag.size() * 2
                                         : align_up(e.draw_mask.size(), 4) * 4 * ((e.K + 31) / 32)), 16);   // one plane of whole quads per 32 isoforms
    d.n_dcls = static_cast<int32_t>(e.dcls_mask.size());
    d.n_units = e.n_units;
    d.max_cls = e.max_cls_size;
    d.n_pairs = static_cast<int32_t>(e.dcls_pairs.size());
    d.off_cls = in_off; in_off = align_up(in_off + e.dcls_tab.size() * 4, 16);
    d.off_clsmask = in_off; in_off = align_up(in_off + e.dcls_pairs.size() * 2, 16);
    d.off_units = in_off; in_off = align_up(in_off + e.unit_desc.size() * 4, 16);
    d.off_sfix = in_off; in_off = align_up(in_off + e.sfix_table.size() * 4, 16);
    d.pe_delta = e.pe_delta ? 1 : 0;
    d.dense_nobad = e.dense_nobad ? 1 : 0;
    d.off_dense = d.off_sfixd = NO_DENSE;
    if (!e.draw_dense.empty()) {
      d.off_dense = in_off; in_off = align_up(in_off + e.draw_dense.size() * 2, 16);
      d.off_sfixd = in_off; in_off = align_up(in_off + e.sfix_dense.size() * 4, 16);
    }
    d.n_mcls = static_cast<int32_t>(e.mcls_tab.size() / (e.K + 1));
    d.off_mcls = in_off; in_off = align_up(in_off + e.mcls_tab.size() * 8, 16);
    d.off_samples = out_off; out_off = align_up(out_off + static_cast<uint64_t>(Sn) * e.K * 8, 16);
    d.off_loglik = out_off; out_off = align_up(out_off + static_cast<uint64_t>(Sn) * 8, 16);
    d.off_drawass = out_off; out_off = align_up(out_off + static_cast<uint64_t>(e.n_draw), 16);
    d.off_stats = out_off; out_off = align_up(out_off + sizeof(ChainStats) * C, 16);
    if (p.want_counts_trace) {
      d.off_trace = out_off;
      out_off = align_up(out_off + static_cast<uint64_t>(M + 1) * C * e.K * 4, 16);
    } else {
      d.off_trace = NO_TRACE;
    }
  }
  in_bytes = std::max<uint64_t>(in_off, 16);
  out_bytes = std::max<uint64_t>(out_off, 16);
  std::vector<unsigned char> h_in(in_bytes, 0);
  for (int i = 0; i < n; i++) {
    const PackedEvent &e = events[i];
    const DevEvent &d = h_events[i];
    std::memcpy(h_in.data() + d.off_consts, e.consts.data(), e.consts.size() * 8);
    std::memcpy(h_in.data() + d.off_base, e.base_count.data(), e.base_count.size() * 4);
    if (e.paired) std::memcpy(h_in.data() + d.off_draw, e.draw_frag.data(), e.draw_frag.size() * 2);
    else {
      // plane w = bits 32 w .. 32 w + 31 of every draw's mask, each plane whole quads long
      uint32_t *pl = reinterpret_cast<uint32_t *>(h_in.data() + d.off_draw);
      const size_t npad = align_up(e.draw_mask.size(), 4), nd = e.draw_mask.size();
      const int W32 = (e.K + 31) / 32, Wx = (e.K + 63) / 64 - 1;
      for (size_t r = 0; r < nd; r++)
        for (int w = 0; w < W32; w++) {
          const uint64_t word = (w < 2) ? e.draw_mask[r] : e.draw_mask_x[r * Wx + (w >> 1) - 1];
          pl[npad * w + r] = static_cast<uint32_t>(word >> (32 * (w & 1)));
        }
    }
    if (!e.unit_desc.empty()) std::memcpy(h_in.data() + d.off_units, e.unit_desc.data(), e.unit_desc.size() * 4);
    if (!e.dcls_tab.empty()) {
      std::memcpy(h_in.data() + d.off_cls, e.dcls_tab.data(), e.dcls_tab.size() * 4);
      if (!e.dcls_pairs.empty())
        std::memcpy(h_in.data() + d.off_clsmask, e.dcls_pairs.data(), e.dcls_pairs.size() * 2);
    }
    if (!e.mcls_tab.empty()) std::memcpy(h_in.data() + d.off_mcls, e.mcls_tab.data(), e.mcls_tab.size() * 8);
    if (!e.sfix_table.empty())
      std::memcpy(h_in.data() + d.off_sfix, e.sfix_table.data(), e.sfix_table.size() * 4);
    if (!e.draw_dense.empty()) {
      std::memcpy(h_in.data() + d.off_dense, e.draw_dense.data(), e.draw_dense.size() * 2);
      std::memcpy(h_in.data() + d.off_sfixd, e.sfix_dense.data(), e.sfix_dense.size() * 4);
    }
  }
  {
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, dev));
    wave_slots = prop.multiProcessorCount * 4 * 2;  // CUs x SIMDs x resident sampler_k2 waves
    if (const char *env = std::getenv("MISO_WAVE_SLOTS")) wave_slots = std::max(64, std::atoi(env));   // experiments: what the planners take as resident
  }
  {
    // The batch's own stream carries its FIRST kernel -- the longest-running class of a whole-gene batch (runs are
    // launched heaviest class first) -- at the highest queue priority; the kernels beside it go to streams of falling
    // priority (launch: stream_for_next), so that whenever a slot frees up the dispatcher takes a workgroup of the longest
    // class still waiting: longest first ACROSS the classes without one kernel holding all their bodies (that kernel was
    // built: 28 minutes of compilation, 800 spilled registers -- tools/experiments/grp_all/).  MEASURED, round 5, and OFF
    // unless MISO_STREAM_PRIO=1: 16 384 genes of 3 - 20 isoforms x 1000 pairs 922 ms with against 917 ms without, hg19-like
    // pair counts 958 against 865 ms (the chains on several workgroups sit in the high-priority kernel and everything else
    // waits behind them) -- profiles/r05_stream_priorities.txt.
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; (void) hipGetLastError(); }
    const char *env = std::getenv("MISO_STREAM_PRIO");
    prio_lo = lo; prio_hi = (env && std::atoi(env) != 0) ? hi : lo;
    if (prio_hi != prio_lo) HIP_OK(hipStreamCreateWithPriority(&stream, hipStreamDefault, prio_hi));
    else HIP_OK(hipStreamCreate(&stream));
  }
  HIP_OK(hipEventCreate(&ev0));
  HIP_OK(hipEventCreate(&ev1));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_events), std::max<size_t>(n, 1) * sizeof(DevEvent)));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_in), in_bytes));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_out), out_bytes));
  if (n) HIP_OK(hipMemcpy(d_events, h_events.data(), n * sizeof(DevEvent), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_in, h_in.data(), in_bytes, hipMemcpyHostToDevice));
  // launch lists: two-isoform events (single- or paired-end) go to sampler_k2, ordered by their
  // number of drawing reads so the chains sharing a wavefront loop equally long
  // (paired-end with a fragment-length distribution so wide that one chain's tables do not fit a CU's LDS -- sd beyond
  // ~400 -- : the two-isoform events take the general kernel, whose tables may stay in global memory, like the
  // reference, which takes any sd, miso_paired.c:299-308; MISO_K2_GENERAL=1 forces it: tests)
  k2_general = p.paired && (48 * fd.prob.size() + 64 > 150 * 1024 || std::getenv("MISO_K2_GENERAL") != nullptr);
  std::vector<int32_t> k2, gen;
  for (int i = 0; i < n; i++) ((events[i].K == 2 && !k2_general) ? k2 : gen).push_back(i);
  // (paired-end: the events sampler_k2's MODE 2 can take come first, MISO_NO_PE_DELTA=1 sends all to MODE 1)
  use_delta = std::getenv("MISO_NO_PE_DELTA") == nullptr;   // fixed here: the slot order depends on it
  // (Round 5 tried ordering a collapsed batch's list by the binomial's likely regime -- inversion below n min(p, q) = 10, BTRS
  // above: a wavefront that holds chains of both runs both -- and measured nothing once the inversion had lost its division:
  // 50.3 - 50.8 ms against 50.1 - 50.6 ms, profiles/r05_sampler_lane_ilp.txt.)
  std::stable_sort(k2.begin(), k2.end(), [&](int x, int y) {
    const bool dx = use_delta && events[x].pe_delta && !events[x].draw_dense.empty(), dy = use_delta && events[y].pe_delta && !events[y].draw_dense.empty();
    return dx != dy ? dx : events[x].n_draw > events[y].n_draw; });
  n_k2w = 0;
  for (int i : k2) n_k2w += (use_delta && events[i].pe_delta && !events[i].draw_dense.empty()) ? 1 : 0;
  // the general kernel's wavefronts loop to their largest K and longest draw list: group alike.
  // Paired-end (sampler_grp gives every chain of a launch the same lanes): genes of very different sizes -- real
  // read counts, 20 ... 10^5 pairs per gene; the reference costs O(reads) per gene, miso_paired.c:393-552 -- are
  // split into size buckets per isoform-count class, each its own launch beside the others.  How many lanes a gene
  // needs follows from its SHARE of the batch's work (drawing quads x isoforms): the batch takes W / (lanes of the
  // device) at best, a chain on L lanes w / L, and a chain should be done in half of that -- L >= 2 x device lanes x
  // w / W.  Up to 1.5 x the rule's 16 lanes: the class's normal launch; up to 64: at least 32 lanes; up to 256: a
  // wavefront of its own (sampler_grp<64, true, KC>); beyond: one chain per workgroup (256 lanes, kernels_grp.inl WIDE
  // -- its four wavefronts each repeat the chain's Metropolis-Hastings step, so only where the read loop dominates);
  // beyond 384: several workgroups (coop.hpp), one per 256 lanes needed.  Never more lanes than the gene has pairs of
  // quads.  Thresholds measured: profiles/r03_pe_buckets.txt.  MISO_NO_PE_BUCKETS=1: one launch per class as before
  // (A/B, tests); MISO_PE_T_WAVE / MISO_PE_T_WIDE: the two thresholds (experiments, tests).
  // (64: 33 ... MISO_MAX_ISOFORMS isoforms -- sampler_wave only, lane k = isoform k; the reference has no limit, miso.c:696)
  // (256: 65 ... MISO_MAX_ISOFORMS isoforms -- sampler_big, the chain's vectors in LDS, kernels_big.hip)
  auto kc_of = [](int K) { return K <= 4 ? 4 : (K <= 8 ? 8 : (K <= 12 ? 12 : (K <= 16 ? 16 : (K <= 32 ? 32 : (K <= 64 ? 64 : 256))))); };
  std::vector<int> bucket(n, 0);   // 0 normal, 1 at least 32 lanes, 2 a wavefront, 3 workgroup-wide (coop_n[event] workgroups)
  coop_n.assign(n, 1);
  if (p.paired && std::getenv("MISO_NO_PE_BUCKETS") == nullptr) {
    double W = 0;
    for (int i : gen) W += static_cast<double>((events[i].n_draw + 3) / 4) * events[i].K;
    const double device_lanes = 2048.0 * 64.0;
    const double share_factor = std::getenv("MISO_PE_SHARE") ? std::atof(std::getenv("MISO_PE_SHARE")) : 2.0;
    const double t_wave = std::getenv("MISO_PE_T_WAVE") ? std::atof(std::getenv("MISO_PE_T_WAVE")) : 64.0;
    const double t_wide = std::getenv("MISO_PE_T_WIDE") ? std::atof(std::getenv("MISO_PE_T_WIDE")) : 256.0;
    const bool dense_ok = std::getenv("MISO_NO_PE_DENSE") == nullptr;
    const bool coop_on = coop_enabled();
    // "at least 32 lanes" from a need of 24 lanes in a batch with a heavy tail (some gene needs a wavefront or more), from 32
    // in a batch of like-sized genes, where a 20-isoform gene of 1000 pairs needs 28 by the share rule and runs better on
    // 16 (round 5, 16 384 genes of 3 - 20 isoforms: 1000 pairs each 920 -> 863 ms with 32, hg19-like pair counts 830 -> 927 ms;
    // profiles/r05_pe_mix_lanes.txt)
    double max_need = 0.0;
    for (int i : gen) max_need = std::max(max_need, W > 0 ? share_factor * device_lanes * (static_cast<double>((events[i].n_draw + 3) / 4) * events[i].K) / W : 0.0);
    const double t_32 = std::getenv("MISO_PE_T_32") ? std::atof(std::getenv("MISO_PE_T_32")) : (max_need > t_wave ? 24.0 : 32.0);
    for (int i : gen) {
      const PackedEvent &e = events[i];
      const int nq = (e.n_draw + 3) / 4;
      const double need = W > 0 ? share_factor * device_lanes * (static_cast<double>(nq) * e.K) / W : 0.0;
      const bool can_wide = dense_ok && !e.draw_dense.empty() && e.K >= 3 && e.K <= PE_DENSE_KMAX;
      if (can_wide && need > t_wide && nq >= 512) {
        bucket[i] = 3;
        if (coop_on && need > 384.0)
          coop_n[i] = std::max(1, std::min({COOP_MAX_N, static_cast<int>(std::ceil(need / 256.0)), nq / 512}));
      } else if (can_wide && need > t_wave && nq >= 128) bucket[i] = 2;
      else if (need > t_32 && nq >= 64) bucket[i] = 1;
    }
    // (round 6) ... and the SMALL genes of a batch of several classes: eight lanes per chain -- a gene of a hundred pairs is all
    // scalar step, which eight chains of a wavefront share instead of four (16 384 genes of 3 - 8 isoforms x 100 / 250 / 500 pairs:
    // 198.5 -> 147.4, 237.3 -> 190.3, 306.2 -> 291.8 ms; profiles/r06_small_genes.txt).  MISO_PE_T_SMALL: drawing quads up to which
    // (0: none; experiments, tests).
    const int t_small = std::getenv("MISO_PE_T_SMALL") ? std::atoi(std::getenv("MISO_PE_T_SMALL")) : 96;
    bool classes2 = false;
    for (int i : gen) classes2 |= kc_of(events[i].K) != kc_of(events[gen[0]].K);
    if (classes2 && t_small > 0 && dense_ok)
      for (int i : gen) {
        const PackedEvent &e = events[i];
        if (bucket[i] == 0 && !e.draw_dense.empty() && e.K >= 3 && e.K <= PE_DENSE_KMAX && (e.n_draw + 3) / 4 <= t_small) bucket[i] = -1;
      }
  }
  std::stable_sort(gen.begin(), gen.end(), [&](int x, int y) {
    const int kx = kc_of(events[x].K), ky = kc_of(events[y].K);
    if (kx != ky) return kx > ky;
    if (bucket[x] != bucket[y]) return bucket[x] > bucket[y];
    return events[x].K != events[y].K ? events[x].K > events[y].K : events[x].n_draw > events[y].n_draw; });
  gen_runs.clear(); tuned_k2_G = 0; k2_plan_key = k2w_plan_key = -1;
  for (size_t j = 0; j < gen.size(); j++) {
    const PackedEvent &e = events[gen[j]];
    const int kc = kc_of(e.K), bk = bucket[gen[j]];
    if (gen_runs.empty() || gen_runs.back().kc != kc || (gen_runs.back().wide ? 3 : (gen_runs.back().wave64 ? 2 : (gen_runs.back().force_G ? 1 : (gen_runs.back().small ? -1 : 0)))) != bk) {
      GenRun r; r.first = static_cast<int>(j); r.kc = kc; r.wide = bk == 3; r.wave64 = bk == 2; r.force_G = bk == 1 ? 32 : (bk == 2 ? 64 : 0); r.small = bk == -1;
      gen_runs.push_back(r);
    }
    GenRun &r = gen_runs.back();
    r.count++;
    r.kmax = std::max(r.kmax, e.K);
    r.kmin = std::min(r.kmin, e.K);
    r.maxq = std::max(r.maxq, (e.n_draw + 3) / 4);
    r.sum_q += (e.n_draw + 3) / 4;
    r.maxcls = std::max(r.maxcls, static_cast<int>(e.dcls_mask.size()));
    if (!e.paired && e.n_draw > 0 && e.dcls_mask.empty()) r.nocls = true;
    if (!e.paired || e.draw_dense.empty() || e.K == 2) r.dense = false;   // (K = 2: draw_dense holds sampler_k2's records)
  }
  n_k2 = static_cast<int>(k2.size()); n_gen = static_cast<int>(gen.size());
  k2.insert(k2.end(), gen.begin(), gen.end());
  h_slots = k2;
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_slots), std::max<size_t>(n, 1) * sizeof(int32_t)));
  if (n) HIP_OK(hipMemcpy(d_slots, k2.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
  if (p.paired) {
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_fp), fd.prob.size() * 8));
    HIP_OK(hipMemcpy(d_fp, fd.prob.data(), fd.prob.size() * 8, hipMemcpyHostToDevice));
  }
  uploaded = true;
  pool_cleared = false; launched_once = false;
}

void miso_batch::launch(uint64_t seed, uint32_t first_event_id) {
  if (!uploaded) MISO_FAIL(MISO_EINVAL, "batch not uploaded");
  HIP_OK(hipSetDevice(device));
  const int n = static_cast<int>(events.size());
  KernelArgs a{};
  a.events = d_events; a.in_pool = d_in; a.out_pool = d_out; a.frag_prob = d_fp;
  a.il = static_cast<int>(fd.prob.size()); a.n_events = n;
  a.C = p.noChains; a.M = p.noIterations; a.B = p.noBurnIn; a.lag = p.noLag;
  a.start = p.start; a.first_event_id = first_event_id; a.seed = seed;
  a.balance = (std::getenv("MISO_PRIO_QUARTILES") && std::atoi(std::getenv("MISO_PRIO_QUARTILES")) != 0) ? 2 : 0;   // experiment: device.hpp prio_by_progress
  a.wide_dedup = (std::getenv("MISO_K2_WIDE_DEDUP") && std::atoi(std::getenv("MISO_K2_WIDE_DEDUP")) == 0) ? 0 : 1;   // kernels_k2.inl: a workgroup-wide chain's Metropolis-Hastings step on four of its eight wavefronts
  a.round_tab = nullptr;
  if (!round_starts.empty()) {   // a later round of stop = CONVERGENT_MEAN (converge_rounds): where the rounds after the first open
    int32_t tab[MISO_MAX_ROUNDS];
    for (int i = 0; i < MISO_MAX_ROUNDS; i++) tab[i] = i < static_cast<int>(round_starts.size()) ? round_starts[i] : -1;
    if (!d_round_tab) HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_round_tab), sizeof(tab)));
    HIP_OK(hipMemcpy(d_round_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
    a.round_tab = d_round_tab;
  }
  last_seed = seed; last_first_event_id = first_event_id;
  converged_done = false;
  if (probe_armed) { HIP_OK(hipStreamSynchronize(probe_stream)); probe_armed = false; }   // (a launch nobody waited for)
  a.pe_force_exact = std::getenv("MISO_K2_SETTLE_ALL") != nullptr;   // tests (kernels_k2.inl: the rescan for high halves on the threshold)
  if (const char *env = std::getenv("MISO_COOP_MAX_POLLS")) a.coop_max_polls = static_cast<uint32_t>(std::max(1L, std::atol(env)));   // tests
  // Trailing sample columns stay 0 (miso.c:661, quirk C8) -- they exist only when the lag does not divide the kept
  // iterations; otherwise the kernels overwrite every sample, log score, pick and statistic of the pool, and clearing
  // 4.8 GB per launch (0.75 ms of the headline's 102 ms step) is for the first launch only.
  if (!pool_cleared || (p.noIterations - p.noBurnIn) % p.noLag != 0 || p.want_counts_trace) {
    HIP_OK(hipMemsetAsync(d_out, 0, out_bytes, stream));
    pool_cleared = true;
  }
  lanes_per_chain = 0;
  last_kernels.clear();
  if (!launched_once) coop_wgs_used = 0;

  // ---- algorithm = MARGINAL / CLASSES (miso.c:272-295, 788-808): Metropolis-Hastings on psi alone, no reads to reassign ----
  // One chain per lane, every event of the batch in one launch (kernels_marginal.hip); its vectors live in LDS,
  // [vector][isoform][lane], 64 lanes per workgroup up to 32 isoforms, 32 beyond.
  if (!p.paired && (p.algorithm == MISO_ALGO_MARGINAL || p.algorithm == MISO_ALGO_CLASSES)) {
    int ks = 2;
    for (const PackedEvent &e : events) ks = std::max(ks, e.K);
    const int lanes = ks <= 32 ? 64 : 32;
    KernelArgs ka = a;
    ka.slot_event = d_slots; ka.n_slots = n; ka.kstride = ks;
    const long chains = static_cast<long>(n) * p.noChains;
    const size_t lds = marginal_lds_bytes(ks, lanes);
    stats_builder = nullptr; kernel_stats.clear();
    HIP_OK(hipEventRecord(ev0, stream));
    if (chains > 0) {
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_marginal), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL(sampler_marginal, dim3(static_cast<unsigned>((chains + lanes - 1) / lanes)), dim3(lanes), lds, stream, ka);
      HIP_OK(hipGetLastError());
    }
    HIP_OK(hipEventRecord(ev1, stream));
    start_clock_probe();
    last_kernels = "sampler_marginal";
    launched = true; launched_once = true; downloaded = false; summarized = false; compared = false;
    return;
  }

  // ---- two-isoform events: sampler_k2<G> ----
  const size_t k2_fp = p.paired ? align_up(fd.prob.size() * 8, 16) : 0;
  const size_t k2_tab = p.paired ? 2 * fd.prob.size() * 4 : 0;   // per chain: int32[2 x il]
  // (only when the batch's wavefronts fit the resident slots once: with several rounds the hardware hands
  // out workgroups as slots free up, heaviest first, and the 4-wavefront workgroups balance better --
  // MISO defaults, 6 chains x 40 000 events at one lane per chain: 374 ms paired, 351 ms unpaired)
  const bool k2_pair_env = !p.paired && !(std::getenv("MISO_K2_PAIR") && std::atoi(std::getenv("MISO_K2_PAIR")) == 0);
  bool k2_pair = k2_pair_env;
  // paired-end: the first n_k2w slots go to MODE 2 (no drawing read with a non-finite score), the rest to
  // MODE 1; single-end: everything is "the rest".  Both keep the event's score table (2 il int32) per chain in LDS.
  // MODE 2 reads the dense records (device.hpp pe_k2_entries): both tables with 2 il + 2 entries
  const size_t k2w_fp = p.paired ? align_up(static_cast<size_t>(pe_k2_entries(static_cast<int>(fd.prob.size()))) * 8, 16) : 0;
  const size_t k2w_tab = p.paired ? static_cast<size_t>(pe_k2_entries(static_cast<int>(fd.prob.size()))) * 4 : 0;
  auto launch_k2 = [&](KernelArgs ka, int G, hipStream_t st, bool wpart = false, int sub_first = 0, int sub_count = -1) {
    int first = wpart ? 0 : n_k2w, count = wpart ? n_k2w : n_k2 - n_k2w;
    if (sub_count >= 0) { first += sub_first; count = sub_count; }   // a part of the list (single-end split, below)
    const long chains = static_cast<long>(count) * p.noChains;
    if (chains <= 0) return;
    ka.slot_event = d_slots + first; ka.n_slots = count;
    const int cpw = 64 / std::max(G, 1);
    const long waves = (chains + cpw - 1) / cpw;
    // single-end: workgroups of 8 wavefronts = one CU's resident slots; the two wavefronts of a SIMD
    // take a heavy and a light group of chains (sampler_k2's WPB = 8).  MISO_K2_PAIR=0: the 4-wavefront
    // workgroups in slot order (round-1 behaviour).
    const bool pair = k2_pair;
    ka.pair_waves = pair ? 1 : 0;
    const unsigned grid = static_cast<unsigned>(pair ? (waves + 7) / 8 : (waves + 3) / 4);
    const size_t k2_lds = (wpart ? k2w_fp : k2_fp) + 4 * static_cast<size_t>(cpw) * (wpart ? k2w_tab : k2_tab);
    if (k2_lds > 160 * 1024) MISO_FAIL(MISO_UNIMPLEMENTED, "Fragment-length distribution too wide for the two-isoform paired-end kernel");
#define MISO_K2_LAUNCH(GG)                                                                              \
  case GG:                                                                                             \
    if (p.paired && wpart) {                                                                           \
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_k2<GG, 2, 4>),                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(k2_lds))); \
      hipLaunchKernelGGL((sampler_k2<GG, 2, 4>), dim3(grid), dim3(256), k2_lds, st, ka);               \
    } else if (p.paired) {                                                                             \
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_k2<GG, 1, 4>),                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(k2_lds))); \
      hipLaunchKernelGGL((sampler_k2<GG, 1, 4>), dim3(grid), dim3(256), k2_lds, st, ka);               \
    } else if (pair) hipLaunchKernelGGL((sampler_k2<GG, 0, 8>), dim3(grid), dim3(512), 0, st, ka);     \
    else hipLaunchKernelGGL((sampler_k2<GG, 0, 4>), dim3(grid), dim3(256), 0, st, ka);                 \
    break;
    switch (G) {
      MISO_K2_LAUNCH(1) MISO_K2_LAUNCH(2) MISO_K2_LAUNCH(3) MISO_K2_LAUNCH(4) MISO_K2_LAUNCH(5)
      MISO_K2_LAUNCH(6) MISO_K2_LAUNCH(7) MISO_K2_LAUNCH(8) MISO_K2_LAUNCH(9) MISO_K2_LAUNCH(10)
      MISO_K2_LAUNCH(12) MISO_K2_LAUNCH(16) MISO_K2_LAUNCH(21) MISO_K2_LAUNCH(32) MISO_K2_LAUNCH(64)
    default: MISO_FAIL(MISO_EINVAL, "MISO_LANES_PER_CHAIN must be one of 1-10,12,16,21,32,64");
    }
#undef MISO_K2_LAUNCH
    HIP_OK(hipGetLastError());
  };

  // ---- more than two isoforms: sampler_grp<G, PE, KC> per isoform-count class ----
  // a workgroup may use half of the CU's 160 KB of LDS (two workgroups per CU)
  // MISO_LDS_MAX_KB (experiments): a larger cap lets one workgroup per CU hold more chains per wavefront
  const size_t LDS_MAX = (std::getenv("MISO_LDS_MAX_KB") ? std::atoi(std::getenv("MISO_LDS_MAX_KB")) : 80) * 1024;
  const size_t fp_plain = p.paired ? align_up(fd.prob.size() * 8, 16) : 0;
  // paired-end dense path (pe_dense): byte records, the probability table with its two extra entries, the
  // score table with rows of il + 2; MISO_NO_PE_DENSE=1 (tests, A/B): the quad loops over the plain records
  const bool dense_env = std::getenv("MISO_NO_PE_DENSE") == nullptr;
  const int il2 = pe_dense_il2(static_cast<int>(fd.prob.size()));
  auto fp_rows = [&](const GenRun &run) { return (p.paired && dense_env && run.dense) ? 1 : 0; };
  auto fp_bytes_of = [&](const GenRun &run) {
    return fp_rows(run) ? align_up(static_cast<size_t>(il2) * 8, 16) : fp_plain;
  };
  struct GrpShape { int qs, ts; };
  auto grp_shape = [&](const GenRun &run) {
    const size_t fp_bytes = fp_bytes_of(run);
    // single-end: per-class thresholds join the slice (class path) when every event has a class
    // table; MISO_NO_CLASS_PATH=1 (tests): force the direct mask path of sampler_grp
    const bool no_cls = std::getenv("MISO_NO_CLASS_PATH") != nullptr;
    int qs = (!p.paired && !no_cls && !run.nocls && run.maxcls > 0) ? run.maxcls : 0;
    if (qs && 4 * 2 * static_cast<size_t>(grp_slice_bytes(run.kmax, qs, 0)) > LDS_MAX) qs = 0;
    // paired-end: the per-event score table (K x il int32) joins the slice when >= 4 chains still fit
    int ts = p.paired ? run.kmax * (fp_rows(run) ? il2 : static_cast<int>(fd.prob.size())) : 0;
    if (ts && fp_bytes + 4 * 4 * static_cast<size_t>(grp_slice_bytes(run.kmax, 0, ts)) > LDS_MAX) ts = 0;
    return GrpShape{qs, ts};
  };
  auto grp_fits = [&](const GenRun &run, const GrpShape &sh, int G) {
    const size_t fp_bytes = fp_bytes_of(run);
    return G >= 2 && G <= 32 && !(G & (G - 1)) &&
           fp_bytes + 4 * static_cast<size_t>(64 / G) * grp_slice_bytes(run.kmax, sh.qs, sh.ts) <= LDS_MAX;
  };
  // Workgroup-wide chains of a run: which chain every workgroup works on (coop.hpp), built once per batch; returns the
  // number of workgroups.
  auto wide_setup = [&](GenRun &run, long chains, hipStream_t st) -> unsigned {
      // Workgroups per chain: what upload() derived from the gene's share of the batch's work (coop_n), at most
      // COOP_MAX_N, all cooperative workgroups of the batch together at most COOP_MAX_WGS (coop.hpp: they must all be
      // resident at once).  MISO_NO_COOP=1: one workgroup per chain; MISO_COOP_DRAWS=n: one per n drawing pairs (tests).
      if (!run.d_coop_tab) {
        run.coop_tab.clear(); run.coop_chains = 0;
        const bool coop_on = coop_enabled();
        const int per_wg = std::getenv("MISO_COOP_DRAWS") ? std::max(256, std::atoi(std::getenv("MISO_COOP_DRAWS"))) : 8192;
        for (long c = 0; c < chains; c++) {
          const int ev_i = h_slots[n_k2 + run.first + c / p.noChains];
          const int nd = events[ev_i].n_draw;
          int nw = !coop_on ? 1 : (std::getenv("MISO_COOP_DRAWS") ? std::min(COOP_MAX_N, std::max(1, (nd + per_wg - 1) / per_wg)) : coop_n[ev_i]);
          if (nw > 1 && coop_wgs_used + nw > coop_gen_budget) nw = std::max(1, coop_gen_budget - coop_wgs_used);
          if (nw > 1) coop_wgs_used += nw;
          for (int r = 0; r < nw; r++) {
            run.coop_tab.push_back(static_cast<int32_t>(c)); run.coop_tab.push_back(r); run.coop_tab.push_back(nw);
            run.coop_tab.push_back(nw > 1 ? run.coop_chains : 0);
          }
          if (nw > 1) run.coop_chains++;
        }
        if (std::getenv("MISO_TIMING")) {
          std::fprintf(stderr, "[miso] wide run kc=%d: %ld chains on %zu workgroups (%d on several:", run.kc, chains, run.coop_tab.size() / 4, run.coop_chains);
          for (size_t i = 0; i < run.coop_tab.size(); i += 4)
            if (run.coop_tab[i + 1] == 0 && run.coop_tab[i + 2] > 1)
              std::fprintf(stderr, " %d x %d draws K=%d", run.coop_tab[i + 2], events[h_slots[n_k2 + run.first + run.coop_tab[i] / p.noChains]].n_draw,
                           events[h_slots[n_k2 + run.first + run.coop_tab[i] / p.noChains]].K);
          std::fprintf(stderr, ")\n");
        }
        HIP_OK(hipMalloc(reinterpret_cast<void **>(&run.d_coop_tab), run.coop_tab.size() * sizeof(int32_t)));
        HIP_OK(hipMemcpy(run.d_coop_tab, run.coop_tab.data(), run.coop_tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(reinterpret_cast<void **>(&run.d_coop_mem), std::max(1, run.coop_chains) * COOP_WORDS * sizeof(uint32_t)));
      }
      HIP_OK(hipMemsetAsync(run.d_coop_mem, 0, std::max(1, run.coop_chains) * COOP_WORDS * sizeof(uint32_t), st));
      return static_cast<unsigned>(run.coop_tab.size() / 4);
  };
  auto launch_grp = [&](KernelArgs ka, GenRun &run, const GrpShape &sh, int G, hipStream_t st) {
    const long chains = static_cast<long>(run.count) * p.noChains;
    ka.slot_event = d_slots + n_k2 + run.first; ka.n_slots = run.count;
    ka.kstride = run.kmax; ka.cstride = sh.qs; ka.tstride = sh.ts;
    const bool wave_kernel = G == 64 && !run.wide && !(run.wave64 && fp_rows(run));   // sampler_wave, not sampler_grp<64, ..>
    ka.pe_dense = wave_kernel ? 0 : fp_rows(run);
    ka.pe_force_exact = std::getenv("MISO_PE_FORCE_EXACT") != nullptr;
    const size_t fp_bytes = wave_kernel ? fp_plain : fp_bytes_of(run);
    if (run.wide) {   // one chain per workgroup: four slices (one per wavefront) + the reduction scratch behind them
      ka.pe_dense = fp_rows(run);
      if (!ka.pe_dense) MISO_FAIL(MISO_EINTERNAL, "workgroup-wide paired-end chains need the dense records");
      const size_t lds0 = align_up(fp_bytes + 4 * static_cast<size_t>(grp_slice_bytes(run.kmax, 0, sh.ts)), 16);
      ka.red_off = static_cast<int32_t>(lds0);
      const size_t lds = lds0 + 96;
      const unsigned grid = wide_setup(run, chains, st);
      ka.coop_tab = run.d_coop_tab; ka.coop_mem = run.d_coop_mem;
#define MISO_GRP_WIDE(KC)                                                                                  \
  {                                                                                                        \
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_grp<64, true, KC, true>),           \
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));        \
    hipLaunchKernelGGL((sampler_grp<64, true, KC, true>), dim3(grid), dim3(256), lds, st, ka);             \
  }
      switch (run.kc) {
      case 4: MISO_GRP_WIDE(4) break;
      case 8: MISO_GRP_WIDE(8) break;
      case 12: MISO_GRP_WIDE(12) break;
      case 16: MISO_GRP_WIDE(16) break;
      default: MISO_GRP_WIDE(32) break;
      }
#undef MISO_GRP_WIDE
    } else if (G == 64 && !wave_kernel) {   // a wavefront per chain, dense records (size bucket between 32 lanes and a workgroup)
      const unsigned grid = static_cast<unsigned>((chains + 3) / 4);
      const size_t lds = fp_bytes + 4 * static_cast<size_t>(grp_slice_bytes(run.kmax, sh.qs, sh.ts));
#define MISO_GRP_WAVE64(KC)                                                                                \
  {                                                                                                        \
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_grp<64, true, KC>),                 \
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));        \
    hipLaunchKernelGGL((sampler_grp<64, true, KC>), dim3(grid), dim3(256), lds, st, ka);                   \
  }
      switch (run.kc) {
      case 4: MISO_GRP_WAVE64(4) break;
      case 8: MISO_GRP_WAVE64(8) break;
      case 12: MISO_GRP_WAVE64(12) break;
      case 16: MISO_GRP_WAVE64(16) break;
      default: MISO_GRP_WAVE64(32) break;
      }
#undef MISO_GRP_WAVE64
    } else if (G == 64 && run.kc > 64) {   // 65 ... MISO_MAX_ISOFORMS isoforms: one wavefront = one workgroup per chain, vectors in LDS (kernels_big.hip)
      const unsigned grid = static_cast<unsigned>(chains);
      const size_t lds = fp_bytes + static_cast<size_t>(run.kmax) * (11 * sizeof(double) + 2 * sizeof(int));
      if (p.paired) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_big<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        hipLaunchKernelGGL(sampler_big<true>, dim3(grid), dim3(64), lds, st, ka);
      } else {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_big<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        hipLaunchKernelGGL(sampler_big<false>, dim3(grid), dim3(64), lds, st, ka);
      }
    } else if (G == 64) {
      const unsigned grid = static_cast<unsigned>((chains + 3) / 4);
      const size_t lds = fp_bytes + 4 * 64 * sizeof(int);
      if (p.paired) hipLaunchKernelGGL(sampler_wave<true>, dim3(grid), dim3(256), lds, st, ka);
      else hipLaunchKernelGGL(sampler_wave<false>, dim3(grid), dim3(256), lds, st, ka);
    } else {
      const int cpw = 64 / G;
      const unsigned grid = static_cast<unsigned>(((chains + cpw - 1) / cpw + 3) / 4);
      const size_t lds = fp_bytes + 4 * static_cast<size_t>(cpw) * grp_slice_bytes(run.kmax, sh.qs, sh.ts);
#define MISO_GRP_LAUNCH_K(GG, KC)                                                                          \
  {                                                                                                        \
    if (p.paired) {                                                                                        \
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_grp<GG, true, KC>),               \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));      \
      hipLaunchKernelGGL((sampler_grp<GG, true, KC>), dim3(grid), dim3(256), lds, st, ka);                 \
    } else {                                                                                               \
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_grp<GG, false, KC>),              \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));      \
      hipLaunchKernelGGL((sampler_grp<GG, false, KC>), dim3(grid), dim3(256), lds, st, ka);                \
    }                                                                                                      \
  }
#define MISO_GRP_LAUNCH(GG)                                    \
  case GG:                                                     \
    switch (run.kc) {                                          \
    case 4: MISO_GRP_LAUNCH_K(GG, 4) break;                    \
    case 8: MISO_GRP_LAUNCH_K(GG, 8) break;                    \
    case 12: MISO_GRP_LAUNCH_K(GG, 12) break;                  \
    case 16: MISO_GRP_LAUNCH_K(GG, 16) break;                  \
    default: MISO_GRP_LAUNCH_K(GG, 32) break;                  \
    }                                                          \
    break;
      switch (G) {
        MISO_GRP_LAUNCH(2) MISO_GRP_LAUNCH(4) MISO_GRP_LAUNCH(8) MISO_GRP_LAUNCH(16) MISO_GRP_LAUNCH(32)
      default: MISO_FAIL(MISO_EINVAL, "MISO_GENERAL_LANES must be 2, 4, 8, 16, 32 or 64");
      }
#undef MISO_GRP_LAUNCH
#undef MISO_GRP_LAUNCH_K
    }
    HIP_OK(hipGetLastError());
  };

  // Lanes per chain, measured: the best split depends on how the batch's wavefronts fill the SIMDs
  // (chains, isoforms, reads, LDS per workgroup), so the first launch of a large single-end batch
  // with more than two isoforms times a few iterations of the rule of thumb's choice and of the next
  // larger ones on the batch itself and keeps the fastest (K=5: 58k -> 81k events/s) (the trial launches record nothing: burn-in = their length; what they leave in the
  // output pool is rewritten by the real launch).  MISO_NO_AUTOTUNE=1 keeps the rule of thumb.
  const bool tune = std::getenv("MISO_NO_AUTOTUNE") == nullptr && p.noIterations >= 2000;   // ~400 trial iterations: <= 20 % of a one-shot run
  auto fastest = [&](const std::vector<int> &cand, auto &&trial) {
    // per-iteration cost = slope between a short and a longer trial (set-up, launch and code-object
    // load cancel); ~200 iterations per candidate
    auto timed = [&](int iters, int g) {
      KernelArgs t = a;
      t.M = iters; t.B = iters;
      float ms = 0;
      HIP_OK(hipEventRecord(ev0, stream));
      trial(t, g);
      HIP_OK(hipEventRecord(ev1, stream));
      HIP_OK(hipEventSynchronize(ev1));
      HIP_OK(hipEventElapsedTime(&ms, ev0, ev1));
      return ms;
    };
    int best = cand[0]; float best_slope = 0;
    for (size_t i = 0; i < cand.size(); i++) {
      timed(4, cand[i]);
      const float t0 = timed(32, cand[i]), t1 = timed(160, cand[i]);
      const float slope = t1 - t0;
      if (i == 0 || slope < best_slope) { best = cand[i]; best_slope = slope; }
    }
    return best;
  };

  // A batch with several kernels (two-isoform events + one general kernel per isoform-count class)
  // runs them CONCURRENTLY, one stream each: every kernel then only has its share of the device to
  // fill, so the lanes-per-chain rule sees `wave_slots x share` (one after the other, each class of
  // a whole-gene batch -- a few thousand chains -- would be spread thin over 32 lanes per chain and
  // still leave the GPU half empty).
  const long total_chains = static_cast<long>(n) * p.noChains;
  const size_t n_kernels = (n_k2 - n_k2w > 0 ? 1 : 0) + (n_k2w > 0 ? 1 : 0) + gen_runs.size();
  auto slots_for = [&](long chains) {
    if (n_kernels <= 1 || total_chains <= 0) return wave_slots;
    return std::max(1, static_cast<int>(static_cast<double>(wave_slots) * chains / total_chains));
  };
  const bool tune_runs = tune && n_kernels <= 1;   // trial launches would time a kernel alone
  int k2_G = 0, k2w_G = 0;
  if (n_k2 - n_k2w > 0) {
    const long chains = static_cast<long>(n_k2 - n_k2w) * p.noChains;
    const int maxq = p.paired ? (events[k2_first_event()].n_draw + 3) / 4 : (events[k2_first_event()].n_draw + 7) / 8;   // Philox blocks
    const int max_cpw = p.paired ? std::max<int>(1, static_cast<int>((60 * 1024 - k2_fp) / (4 * k2_tab))) : 64;
    if (const char *env = std::getenv("MISO_LANES_PER_CHAIN")) k2_G = std::atoi(env);
    else if (tuned_k2_G) k2_G = tuned_k2_G;
    else {
      k2_G = choose_lanes_per_chain(chains, maxq, slots_for(chains), max_cpw);
      if (tune_runs && chains >= 4096 && std::getenv("MISO_AUTOTUNE_K2") != nullptr) {   // the rule is the measured optimum
        static const int kG[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 21, 32, 64};
        std::vector<int> cand{k2_G};
        for (int i = 0; i < 15; i++)
          if (kG[i] == k2_G) {
            for (int j : {i - 1, i + 1, i + 2})
              if (j >= 0 && j < 15 && 64 / kG[j] <= max_cpw && !(p.paired && (kG[j] & (kG[j] - 1))) &&
                  kG[j] <= std::max(1, maxq))
                cand.push_back(kG[j]);
          }
        if (cand.size() > 1) k2_G = fastest(cand, [&](const KernelArgs &t, int g) { launch_k2(t, g, stream); });
        tuned_k2_G = k2_G;
      }
    }
  }
  if (n_k2w > 0) {   // MODE 2: two workgroups per CU may share the LDS (80 KB each)
    const long chains = static_cast<long>(n_k2w) * p.noChains;
    const int maxq = (events[k2_first_event()].n_draw + 3) / 4;
    const long lds_left = static_cast<long>(LDS_MAX) - static_cast<long>(k2w_fp);
    const int max_cpw = static_cast<int>(std::max<long>(1, lds_left / static_cast<long>(4 * k2w_tab)));
    if (const char *env = std::getenv("MISO_LANES_PER_CHAIN")) {
      k2w_G = std::atoi(env);
      if (k2w_G < 1 || k2w_G > 64) MISO_FAIL(MISO_EINVAL, "MISO_LANES_PER_CHAIN must be one of 1-10,12,16,21,32,64");
    } else k2w_G = choose_lanes_per_chain(chains, maxq, slots_for(chains), max_cpw);
    while (k2w_G < 64 && static_cast<size_t>(64 / k2w_G) > static_cast<size_t>(max_cpw)) k2w_G *= 2;   // a forced choice never exceeds the LDS
  }
  // single-end runs whose events all have a class table go to sampler_flat (kernels_flat.inl): NC chains
  // per wavefront, as many as fit the wavefront's share of the LDS (two workgroups of four wavefronts
  // per CU: 20 KB each), fewer when the batch would not fill the device otherwise.
  // MISO_NO_FLAT=1: sampler_grp as in round 1 (A/B, tests); MISO_FLAT_NC=n forces the chains per wavefront.
  // collapsed single-end batches (miso_batch_set_collapsed): the events with three and more isoforms in ONE launch of
  // sampler_lane_k (kernels_lane.hip), provided every one of them has its class table
  // (level 2 only: with the classes of a five- or ten-isoform event sharing ~1000 reads the chains of small binomials
  // cost more than the read sweep they replace -- profiles/r03_collapsed.txt; it pays from ~10^4 reads per event)
  // The route is decided per run: a class whose events all have their class tables takes sampler_lane_k, a
  // run without them -- or of 33 isoforms and more: several mask words, sampler_wave / sampler_big only -- keeps the
  // per-read kernels: its events are sampled per read, i.e. equal the CPU checker's COUNTER mode
  // (tests/test_gpu_collapsed.py).
  std::vector<char> run_lane(gen_runs.size(), 0);
  bool lane_gen = false;
  if (collapsed && collapsed_level >= 2 && !p.paired && n_gen > 0)
    for (size_t ri = 0; ri < gen_runs.size(); ri++) {
      run_lane[ri] = !(gen_runs[ri].nocls || gen_runs[ri].kc >= 64);
      lane_gen |= run_lane[ri] != 0;
    }
  std::vector<int> flat_nc(gen_runs.size(), 0), flat_nc_max(gen_runs.size(), 0);
  for (size_t ri = 0; ri < gen_runs.size(); ri++) {
    const GenRun &run = gen_runs[ri];
    if (run_lane[ri]) continue;
    if (p.paired || run.nocls || run.kc >= 64 || std::getenv("MISO_NO_FLAT") != nullptr) continue;
    const int slice = flat_layout(run.kmax, std::max(run.maxcls, 1)).bytes;
    // workgroups per CU the chains per wavefront are sized for: kernels_flat.inl's register budgets -- 3 up to four isoforms
    // and for nine to twelve (measured round 4, profiles/r04_occupancy.txt), 2 otherwise (five to eight isoforms: the kernel
    // allows 3, sizing for 3 gained nothing); MISO_FLAT_WGS overrides
    const int wgs = std::getenv("MISO_FLAT_WGS") ? std::max(1, std::atoi(std::getenv("MISO_FLAT_WGS"))) : flat_wgs_for(run.kc);
    const size_t lds_wave = std::getenv("MISO_LDS_MAX_KB") ? LDS_MAX / 4 : static_cast<size_t>(160 * 1024 / (4 * wgs)) - 64;
    const int nc_max = std::min<int>(64, static_cast<int>(lds_wave) / slice);
    if (nc_max < 1) continue;
    const long chains = static_cast<long>(run.count) * p.noChains;
    // the fewest rounds of resident wavefronts that nc_max allows, then the fewest chains per wavefront
    // that still makes that many rounds (a 40 000-chain batch at nc_max = 17 would run 1.15 rounds = 2)
    const long slots = std::max<long>(1, static_cast<long>(slots_for(chains)) * wgs / 2);
    const long rounds = std::max<long>(1, (chains + slots * nc_max - 1) / (slots * nc_max));
    int nc = static_cast<int>(std::min<long>(nc_max, std::max<long>(1, (chains + slots * rounds - 1) / (slots * rounds))));
    // five to eight isoforms: how the 64 lanes divide over a wavefront's chains and items matters more than the
    // rounds -- measured on one device, 40 000 events (events/s at 5 / 6 / 7 / 8 / 10 chains per wavefront):
    // K=6 69.4k / 67.5k / 72.2k / 77.2k / 68.6k, K=7 64.1k / 61.4k / 68.8k / 73.5k / 64.3k, K=8 55.8k / 57.1k /
    // 52.6k / 55.1k / 54.4k; K=5: 8 -> 93.4k, 7 -> 90.3k, 10 -> 86.8k.  (Timing the candidates on the first launch
    // was tried: a 7 % difference is inside the noise of a 200-iteration trial, the choice flipped between runs.)
    if (run.kc == 8) nc = std::min(nc, run.kmin >= 8 ? 6 : 8);
    if (run.kc == 4 && run.kmin >= 4) nc = std::min(nc, 8);   // four isoforms, descriptor loop: 8 -> 119.8k, 16 -> 116.1k, 13 -> 85.5k
    if (const char *env = std::getenv("MISO_FLAT_NC")) nc = std::max(1, std::min(nc_max, std::atoi(env)));
    flat_nc[ri] = nc; flat_nc_max[ri] = nc_max;
  }
  // up to twenty isoforms the kernel of the launch's largest isoform count (its slice layout at compile time,
  // kernels_flat.inl); MISO_FLAT_NO_KS=1: the run-time layout everywhere (A/B, tests)
  auto flat_ks = [&](const GenRun &run) {
    const int lo = run.kc == 4 ? 3 : (run.kc == 8 ? 5 : (run.kc == 12 ? 9 : (run.kc == 16 ? 13 : 17)));
    return (run.kc <= 32 && run.kmax >= lo && run.kmax <= std::min(run.kc, 20) && std::getenv("MISO_FLAT_NO_KS") == nullptr) ? run.kmax : 0;
  };
  // ... and, when every event of the launch has that count, the kernel that knows it (UNI: no per-chain `k < K` masks)
  auto flat_uni = [&](const GenRun &run) { return flat_ks(run) > 0 && run.kmin == run.kmax && std::getenv("MISO_FLAT_NO_UNI") == nullptr; };
  auto flat_name = [&](const GenRun &run) {
    return "sampler_flat<" + std::to_string(run.kc) + ", " + std::to_string(flat_ks(run)) + (flat_uni(run) ? ", true>" : ">");
  };
  // Which chains a wavefront of sampler_flat owns (kernels_flat.inl: a.wave_tab).  Uniform batches: `nc` consecutive
  // chains each.  When the batch's events differ widely in size -- the heaviest wavefront of the uniform rule would
  // carry more than twice the average wavefront's work units -- the wavefronts are packed by UNITS instead: as many
  // consecutive chains (at most nc_max: the LDS) as make about the average wavefront's units, a chain of several
  // times that alone in a WORKGROUP (FLAT_WIDE: 256 lanes), workgroups ordered heaviest first (the hardware starts
  // them in index order).  MISO_FLAT_PACK=0 / 1: never / always (tests, A/B).
  // How many units per wavefront: as many as nc_max average chains have -- the scalar step costs a wavefront the same
  // for 2 chains and for 13, so fewer, fuller wavefronts win (hg19-like read counts, 40 000 events, K = 5: 497 ms at 8
  // average chains per wavefront, 425 ms at 13+) -- unless that leaves the device part empty: fewer wavefronts than
  // resident slots, or a second round that is less than half full (K = 3: 3349 wavefronts on 3072 slots 286 ms,
  // 5704 wavefronts 270 ms); then the wavefront count goes to 0.95 x one or two rounds
  // (profiles/r03_flat_pack_sweep.txt).  A forced MISO_FLAT_NC sets the average chains per wavefront instead.
  auto flat_waves = [&](GenRun &run, int nc, int nc_max_u, long resident_u, int nc_max_p, long resident_p) {
    const long chains = static_cast<long>(run.count) * p.noChains;
    const int C = p.noChains;
    const char *env = std::getenv("MISO_FLAT_PACK");
    const long key = (((static_cast<long>(nc) * 4 + (env ? 1 + (std::atoi(env) != 0) : 0)) * 128 + nc_max_u) * 128 + nc_max_p) * 1024 + (std::getenv("MISO_FLAT_PACK_OV") ? 1 + std::atol(std::getenv("MISO_FLAT_PACK_OV")) % 1000 : 0);
    if (run.wave_key == key && run.d_wave_tab) return;
    // (round 5) what a chain costs its wavefront, in work units: its units + the scalar step and thresholds, which do not
    // depend on the reads.  Measured per chain-iteration at eight / five chains per wavefront (profiles/r05_flat_chunks.txt):
    // K = 5 238 VALU against 1.33 per unit, K = 10 641 against 1.96, i.e. about 30 K + 25 units -- but the wavefronts this
    // matters for carry 14 - 18 small chains, whose flat passes and leader sections are shared by twice as many chains:
    // 16 K + 14.  Swept on hg19-like read counts at 25 ... 300 % of the first figure: K = 3 / 4 / 6 / 8 best at 50 % (208 /
    // 152 / 107 / 80 k events/s against 153 / 144 / 97 / 74 k at 100 %), K = 5 / 12 at 100 % (122 / 45 k against 114 / 43 k);
    // by units alone: 151 / 112 / 55 k at K = 3 / 5 / 10.  MISO_FLAT_PACK_OV=<per cent of the rule> (0: by units alone).
    const char *ov_env = std::getenv("MISO_FLAT_PACK_OV");
    const long ov_pct = ov_env ? std::atol(ov_env) : 100;   // (per cent of the rule: experiments)
    auto units_of = [&](long c) {
      const PackedEvent &e = events[h_slots[n_k2 + run.first + c / C]];
      return static_cast<long>(e.n_units) + (16L * e.K + 14) * ov_pct / 100;
    };
    long total = 0, head = 0;
    for (long c = 0; c < chains; c++) { const long u = units_of(c); total += u; if (c < nc) head += u; }
    long heaviest = head;   // the list is ordered by isoforms first: look at every wavefront of the uniform rule
    for (long c = 0, sum = 0; c < chains; c++) {
      sum += units_of(c);
      if ((c + 1) % nc == 0 || c + 1 == chains) { heaviest = std::max(heaviest, sum); sum = 0; }
    }
    const double mean_wave = static_cast<double>(total) * nc / std::max<long>(1, chains);
    const bool pack = env ? std::atoi(env) != 0 : (chains > nc && static_cast<double>(heaviest) > 2.0 * mean_wave);
    struct W { int32_t first, n; long units; };
    std::vector<W> waves, wides;
    if (!pack) {
      for (long c = 0; c < chains; c += nc) waves.push_back(W{static_cast<int32_t>(c), static_cast<int32_t>(std::min<long>(nc, chains - c)), 0});
      run.wave_nc = nc;
    } else {
      // (packed launches have their own sizing: launch_flat)
      const int nc_max = nc_max_p; const long resident = resident_p;
      const int cap = std::min(nc_max, 255);
      int most = 1;
      auto pack_with = [&](double U) {
        waves.clear(); wides.clear(); most = 1;
        const double wide_min = std::max(3.0 * U, 2048.0);
        for (long c = 0; c < chains;) {
          const long u = units_of(c);
          if (static_cast<double>(u) >= wide_min) { wides.push_back(W{static_cast<int32_t>(c), 1, u}); c++; continue; }
          W w{static_cast<int32_t>(c), 0, 0};
          while (c < chains && w.n < cap) {
            const long v = units_of(c);
            if (static_cast<double>(v) >= wide_min || (w.n > 0 && static_cast<double>(w.units + v) > 1.05 * U)) break;
            w.units += v; w.n++; c++;
          }
          most = std::max(most, static_cast<int>(w.n));
          waves.push_back(w);
        }
        return static_cast<long>(waves.size() + 4 * wides.size());
      };
      const double per_chain = static_cast<double>(total) / std::max<long>(1, chains);
      const bool forced = std::getenv("MISO_FLAT_NC") != nullptr;
      const long n_waves = pack_with(std::max(64.0, per_chain * (forced ? nc : cap)));
      if (!forced && n_waves < 3 * resident / 2) {
        // (round 6) ... and TWO rounds are aimed at from further below than one: the second round's workgroups start as the
        // first round's end, staggered, and a count within a few per cent of two full rounds runs into a third (hg19-like read
        // counts, K = 5, 2048 resident wavefronts: 4029 wavefronts 71.7 ms, 3793 64.6 ms, 3541 66.2 ms, 2956 -- the fullest
        // packing -- 67.7 ms; profiles/r06_flat_pack_rounds.txt).  MISO_FLAT_ROUNDS_FRAC: the fraction of two rounds (experiments).
        const bool two = n_waves > resident;
        const double frac2 = std::getenv("MISO_FLAT_ROUNDS_FRAC") ? std::atof(std::getenv("MISO_FLAT_ROUNDS_FRAC")) : 0.85;
        const double want = (two ? frac2 : 0.95) * static_cast<double>(two ? 2 * resident : resident);
        if (static_cast<double>(n_waves) < want) {
          // (round 5) ... and not a few wavefronts MORE than the round(s): the packing's bound is a target, the count it
          // makes lands some per cent off, and 3110 wavefronts on 3072 slots cost a second round for 38 of them
          // (K = 3, hg19-like read counts: 154 k events/s there, 212 k at 2960; profiles/r05_flat_chunks.txt)
          const long target = two ? static_cast<long>((frac2 + 0.04) * 2.0 * static_cast<double>(resident)) : resident;
          double U2 = std::max(64.0, static_cast<double>(total) / want);
          long n2 = pack_with(U2);
          for (int it = 0; it < 8 && n2 > target; it++) {
            const long before = n2;
            U2 *= static_cast<double>(n2) / (0.97 * static_cast<double>(target));
            n2 = pack_with(U2);
            if (n2 >= before) break;   // the chains per wavefront are at the LDS's limit: fuller wavefronts are not to be had
          }
        }
      }
      if (std::getenv("MISO_TIMING"))
        std::fprintf(stderr, "[flat_waves] kc %d chains %ld cost/chain %.1f cap %d resident %ld first pack %ld waves -> %zu waves + %zu wide, most chains %d\n",
                     run.kc, chains, per_chain, cap, resident, n_waves, waves.size(), wides.size(), most);
      std::stable_sort(wides.begin(), wides.end(), [](const W &x, const W &y) { return x.units > y.units; });
      std::stable_sort(waves.begin(), waves.end(), [](const W &x, const W &y) { return x.units > y.units; });
      run.wave_nc = most;
    }
    run.wave_tab.clear();
    for (const W &w : wides) for (int i = 0; i < 4; i++) { run.wave_tab.push_back(w.first); run.wave_tab.push_back(1 | FLAT_WIDE); }
    for (const W &w : waves) { run.wave_tab.push_back(w.first); run.wave_tab.push_back(w.n); }
    while ((run.wave_tab.size() / 2) % 4) { run.wave_tab.push_back(0); run.wave_tab.push_back(0); }   // padding wavefronts
    run.wave_wide = static_cast<int>(wides.size());
    run.wave_packed = pack;
    if (run.d_wave_tab) (void) hipFree(run.d_wave_tab);
    run.d_wave_tab = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&run.d_wave_tab), std::max<size_t>(run.wave_tab.size(), 2) * sizeof(int32_t)));
    HIP_OK(hipMemcpy(run.d_wave_tab, run.wave_tab.data(), run.wave_tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    run.wave_key = key;
  };
  auto launch_flat = [&](KernelArgs ka, GenRun &run, int nc, int nc_max, hipStream_t st) {
    {
      const long chains = static_cast<long>(run.count) * p.noChains;
      const int wgs = std::getenv("MISO_FLAT_WGS") ? std::max(1, std::atoi(std::getenv("MISO_FLAT_WGS"))) : flat_wgs_for(run.kc);
      // Five to eight isoforms: the kernel's registers allow three workgroups per CU; launches of like-sized events are sized for
      // two (fuller wavefronts: K = 6 312 against 333 ms, K = 7 347 / 366), launches packed by cost for three -- more, smaller
      // wavefronts level the heavy tail better than fuller ones amortise the scalar step (hg19-like read counts, 40 000 events:
      // K = 5 276 -> 252 ms, K = 6 332 -> 278, K = 7 379 -> 339, K = 8 415 -> 393; profiles/r06_flat_licm.txt)
      int wgs_p = wgs, nc_max_p = nc_max;
      if (run.kc == 8 && std::getenv("MISO_FLAT_WGS") == nullptr && std::getenv("MISO_LDS_MAX_KB") == nullptr && std::getenv("MISO_FLAT_PACK_WGS2") == nullptr) {
        wgs_p = 3;
        const int slice = flat_layout(run.kmax, std::max(run.maxcls, 1)).bytes;
        nc_max_p = std::max(1, std::min<int>(64, static_cast<int>(160 * 1024 / (4 * wgs_p) - 64) / slice));
      }
      flat_waves(run, nc, nc_max, std::max<long>(1, static_cast<long>(slots_for(chains)) * wgs / 2),
                 nc_max_p, std::max<long>(1, static_cast<long>(slots_for(chains)) * wgs_p / 2));
    }
    ka.slot_event = d_slots + n_k2 + run.first; ka.n_slots = run.count;
    ka.kstride = run.kmax; ka.cstride = std::max(run.maxcls, 1); ka.tstride = 0; ka.nc = run.wave_nc;
    ka.wave_tab = run.d_wave_tab;
    // the descriptor read loop from four isoforms on (three: the walking loop is 2 % faster -- two thresholds per unit,
    // little to save) and whenever a chain owns a whole workgroup; MISO_FLAT_NO_DESC=1: the walking loop everywhere
    // (A/B, tests; not with workgroup-wide chains)
    ka.flat_desc = ((std::getenv("MISO_FLAT_NO_DESC") == nullptr && run.kmax >= 4) || run.wave_wide > 0) ? 1 : 0;
    // Thresholds only for the chains whose psi changed (kernels_flat.inl): the pass is one lane per (chain, class) and the
    // Metropolis-Hastings step rejects 45 - 65 % of the proposals, but numbering the chains that accepted costs every lane
    // a few instructions -- it pays where the thresholds weigh enough: from six isoforms on (K = 6 ... 12: + 4.5 ... 6.4 %)
    // and for small events (hg19-like read counts at K = 5: + 5.5 %; 1000 reads per event at K = 3 ... 5: - 1 ... 2 %,
    // profiles/r04_occupancy.txt).  MISO_FLAT_THR_SKIP=0 / 1: never / always (A/B, tests).
    {
      double units = 0;
      for (int j = 0; j < run.count; j++) units += events[h_slots[n_k2 + run.first + j]].n_units;
      const char *env = std::getenv("MISO_FLAT_THR_SKIP");
      ka.flat_thr_skip = env ? (std::atoi(env) != 0) : (run.kmax >= 6 || units < 150.0 * run.count);
    }
    const unsigned grid = static_cast<unsigned>(run.wave_tab.size() / 8);
    const size_t lds = 4 * static_cast<size_t>(run.wave_nc) * flat_layout(ka.kstride, ka.cstride).bytes;
    // Priority by progress (device.hpp prio_by_progress) when the launch takes several rounds of like-sized wavefronts:
    // 40 000 events x 1000 reads, K = 5 348.0 -> 334.4 ms, K = 10 649.4 -> 631.4 ms, same box; wavefronts packed by cost
    // (hg19-like read counts) lose 1 % and keep the arbiter's own order; so do the paired-end and the two-isoform kernels
    // (+ 0.5 ... 4 %: profiles/r06_prio_by_progress.txt).  MISO_PRIO_QUARTILES=0 / 1: never / everywhere (A/B).
    if (std::getenv("MISO_PRIO_QUARTILES") == nullptr && !run.wave_packed && run.wave_wide == 0 &&
        static_cast<long>(grid) * 4 > static_cast<long>(slots_for(static_cast<long>(run.count) * p.noChains)) * flat_wgs_for(run.kc) / 2)
      ka.balance = 2;
#define MISO_FLAT_LAUNCH_(KC, KS, UNI)                                                                  \
  {                                                                                                     \
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_flat<KC, KS, UNI>),              \
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));     \
    hipLaunchKernelGGL((sampler_flat<KC, KS, UNI>), dim3(grid), dim3(256), lds, st, ka);                \
  }
#define MISO_FLAT_LAUNCH(KC, KS)                                                                        \
  {                                                                                                     \
    if (KS > 0 && flat_uni(run)) MISO_FLAT_LAUNCH_(KC, KS, (KS > 0)) else MISO_FLAT_LAUNCH_(KC, KS, false)  \
  }
    const int ksel = flat_ks(run);   // (== ka.kstride or 0)
    switch (run.kc) {
    case 4:
      if (ksel == 3) MISO_FLAT_LAUNCH(4, 3) else if (ksel == 4) MISO_FLAT_LAUNCH(4, 4) else MISO_FLAT_LAUNCH(4, 0)
      break;
    case 8:
      if (ksel == 5) MISO_FLAT_LAUNCH(8, 5) else if (ksel == 6) MISO_FLAT_LAUNCH(8, 6) else if (ksel == 7) MISO_FLAT_LAUNCH(8, 7)
      else if (ksel == 8) MISO_FLAT_LAUNCH(8, 8) else MISO_FLAT_LAUNCH(8, 0)
      break;
    case 12:
      if (ksel == 9) MISO_FLAT_LAUNCH(12, 9) else if (ksel == 10) MISO_FLAT_LAUNCH(12, 10) else if (ksel == 11) MISO_FLAT_LAUNCH(12, 11)
      else if (ksel == 12) MISO_FLAT_LAUNCH(12, 12) else MISO_FLAT_LAUNCH(12, 0)
      break;
    case 16:
      if (ksel == 13) MISO_FLAT_LAUNCH(16, 13) else if (ksel == 14) MISO_FLAT_LAUNCH(16, 14) else if (ksel == 15) MISO_FLAT_LAUNCH(16, 15)
      else if (ksel == 16) MISO_FLAT_LAUNCH(16, 16) else MISO_FLAT_LAUNCH(16, 0)
      break;
    default:
      if (ksel == 17) MISO_FLAT_LAUNCH(32, 17) else if (ksel == 18) MISO_FLAT_LAUNCH(32, 18) else if (ksel == 19) MISO_FLAT_LAUNCH(32, 19)
      else if (ksel == 20) MISO_FLAT_LAUNCH(32, 20) else MISO_FLAT_LAUNCH(32, 0)
      break;
    }
#undef MISO_FLAT_LAUNCH
#undef MISO_FLAT_LAUNCH_
    HIP_OK(hipGetLastError());
  };
  std::vector<int> grp_G(gen_runs.size(), 64);
  std::vector<GrpShape> grp_sh(gen_runs.size());
  for (size_t ri = 0; ri < gen_runs.size(); ri++) {
    GenRun &run = gen_runs[ri];
    if (run_lane[ri]) continue;
    if (run.kc >= 64) { flat_nc[ri] = 0; grp_G[ri] = 64; grp_sh[ri] = GrpShape{0, 0}; continue; }   // more than 32 isoforms: one wavefront per chain
    // sampler_flat or sampler_grp?  Measured on the batch's first launch like the lanes per chain below
    // (flat wins at every isoform count of profiles/r02_flat_vs_grp_sweep.txt but 5); a small or untuned
    // batch takes sampler_flat.
    const bool contest = flat_nc[ri] > 0 && tune_runs && static_cast<long>(run.count) * p.noChains >= 2048 &&
                         run.tuned_flat < 0 && std::getenv("MISO_FLAT_NC") == nullptr;
    if (flat_nc[ri] > 0 && run.tuned_flat == 0) flat_nc[ri] = 0;
    if (flat_nc[ri] > 0 && !contest) continue;
    GrpShape sh = grp_sh[ri] = grp_shape(run);
    const long chains = static_cast<long>(run.count) * p.noChains;
    int G = 64;
    if (run.wide) {   // one chain per workgroup (launch_grp): the score table joins the slice when four of them fit
      GrpShape w{0, 0};
      w.ts = run.kmax * il2;
      if (fp_bytes_of(run) + 4 * static_cast<size_t>(grp_slice_bytes(run.kmax, 0, w.ts)) + 64 > LDS_MAX) w.ts = 0;
      grp_sh[ri] = w;
      grp_G[ri] = 64;
      continue;
    }
    if (const char *env = std::getenv("MISO_GENERAL_LANES")) {
      G = std::atoi(env);
    } else if (const char *envc = std::getenv("MISO_GENERAL_LANES_BY_CLASS")) {   // experiments: "4:16,8:16,12:32,16:32,32:32"
      G = 16;
      for (const char *q = envc; q && *q;) {
        int kc = 0, g = 0;
        if (std::sscanf(q, "%d:%d", &kc, &g) == 2 && kc == run.kc) G = g;
        q = std::strchr(q, ',');
        if (q) q++;
      }
    } else if (run.tuned_G) {
      G = run.tuned_G;
      if (p.paired && sh.ts && !grp_fits(run, sh, G)) {   // chosen with the score tables in global memory (the eight-lane rule below)
        GrpShape s0 = sh;
        s0.ts = 0;
        if (grp_fits(run, s0, G)) sh = grp_sh[ri] = s0;
      }
    } else {
      // rule of thumb.  single-end: the per-iteration scalar step costs the same per wavefront
      // whatever G is, so pack as many chains per wavefront as still fills the device: the smallest
      // G whose wavefronts occupy every resident slot; a batch too small for that takes the largest.
      // paired-end: the read loop waits on memory and more wavefronts hide it: largest G up to 16; 32 from
      // nine isoforms on (measured, pe_dense: K=10 19.5k -> 19.9k, K=20 7.35k -> 7.57k events/s; K=5 29.9k ->
      // 28.9k, K=3 53k -> 46k) and whenever 16 would leave the kernel's share of the device unfilled.
      bool found = false;
      // (a whole-gene mix -- several classes side by side -- runs 9 % faster with 32 everywhere: 14.4k -> 15.7k genes/s)
      // (round 3, 20 000 events of one isoform count: 16 lanes beat 32 from nine isoforms on as well once the launch
      // is more than a round and a half of wavefronts -- K = 9 21.3k -> 26.8k, K = 10 20.8k -> 26.9k, K = 12 19.1k ->
      // 22.5k, K = 16 14.4k -> 16.6k, K = 20 10.7k -> 10.9k events/s; the mix keeps 32: 16.8k vs 14.2k genes/s;
      // profiles/r03_pe_lanes_sweep.txt)
      int n_classes = 0;
      for (size_t rj = 0; rj < gen_runs.size(); rj++) if (rj == 0 || gen_runs[rj].kc != gen_runs[rj - 1].kc) n_classes++;
      // (round 3, 8 hardware queues, 16 384 genes of 3-20 isoforms per 1500 iterations: 16 lanes up to 16 isoforms against 32
      // everywhere -- 400 pairs per gene 120 vs 150 ms, 1000 pairs 194 vs 206 ms, hg19-like counts 176 vs 175 ms, 2500 pairs
      // 436 vs 367 ms: small genes share the scalar step four to a wavefront, large ones want the lanes; the switch sits at a
      // mean of 400 drawing quads per gene of the class; MISO_PE_MIX_LANES=16|32 forces)
      const double mean_q = run.sum_q / std::max(1, run.count);
      const char *mixl = std::getenv("MISO_PE_MIX_LANES");
      // (round 5, same batch on one box, 7500 iterations: the 17 - 20 isoform class on 16 lanes as well -- 1000 pairs per gene
      // 930 -> 852 ms, hg19-like pair counts 865 -> 820 ms: four chains of a wavefront share the scalar step, and the class
      // no longer fills the device with half-efficient wavefronts for the launch's first third while the other classes
      // wait; the bucket of "at least 32 lanes" from a need of 32 lanes instead of 24: a 20-isoform gene of 1000 pairs
      // needs 28 by the share rule; profiles/r05_pe_mix_lanes.txt)
      const bool mix16 = run.kc <= 32 && !run.force_G && (mixl ? std::atoi(mixl) == 16 : mean_q < 400.0);
      const bool pe32 = p.paired && ((n_classes + (n_k2 > 0 ? 1 : 0) > 1 && !mix16) || (run.kc >= 12 && 2 * ((chains + 3) / 4) < 3 * static_cast<long>(slots_for(chains))) ||
                                     (chains + 3) / 4 < slots_for(chains));
      for (int g : {2, 4, 8, 16, 32}) {
        if (p.paired && g > 16 && found && !pe32) break;
        if (!grp_fits(run, sh, g)) continue;
        const int cpw = 64 / g;
        if (!found || g <= std::max(2, run.maxq)) G = g;
        found = true;
        if (!p.paired && (chains + cpw - 1) / cpw >= slots_for(chains)) break;
      }
      // Round 5, paired-end, ONE isoform-count class in the batch, five to nine isoforms, two rounds of wavefronts and more at
      // eight lanes: EIGHT lanes per chain -- eight chains of a wavefront share the scalar step instead of four.  Rounds 2 - 4
      // measured that slower (the read loop waited on memory and more wavefronts hid it); since the draws are ordered by
      // fragment rows and the loop's header no longer waits for fresh gathers it wins: 40 000 events x 1000 pairs, K = 5 / 6 / 7 /
      // 8 / 9: 45.3 -> 54.2, 41.5 -> 48.7, 40.5 -> 45.5, 38.0 -> 40.5, 30.2 -> 31.5 k events/s; K = 12: 25.7 -> 21.2 k, four lanes
      // 33.7 k at K = 5; in a whole-gene mix, where a class is a few thousand genes, 19.2 -> 18.2 k genes/s: not there
      // (profiles/r05_lanes_sweep.txt).  MISO_PE_LANES8=0 / 1: never / also in a mix and for the size buckets' normal run.
      {
        const char *l8 = std::getenv("MISO_PE_LANES8");
        const bool single = n_classes + (n_k2 > 0 ? 1 : 0) == 1;
        const bool want8 = l8 && std::atoi(l8) != 2 ? std::atoi(l8) != 0 : ((single || l8) && chains >= 8L * 2 * slots_for(chains));   // (2: the rule in a mix of classes too)
        if (p.paired && want8 && G == 16 && !run.force_G && run.kmin >= 5 && run.kmax <= 9 && grp_fits(run, sh, 8)) G = 8;
        // Three and four isoforms the same -- there the class keeps its score tables in LDS (grp_shape), which holds four
        // chains per wavefront and no more: eight chains with the tables in global memory (an L2 line per pick) instead,
        // K = 3 66.6 -> 78.9 k, K = 4 58.1 -> 68.0 k events/s; the tables alone are worth 2 % (16 lanes without them 65.3 /
        // 57.0 k), four lanes lose (55.9 / 37.6 k) (profiles/r05_lanes_sweep.txt).
        if (p.paired && want8 && G == 16 && !run.force_G && run.kmax <= 4) {
          GrpShape sh8 = sh;
          if (!grp_fits(run, sh8, 8)) sh8.ts = 0;
          if (grp_fits(run, sh8, 8)) { sh = grp_sh[ri] = sh8; G = 8; }
        }
      }
      if (tune_runs && chains >= 2048 && G != 64) {
        std::vector<int> cand{G};
        // the rule errs on the small side (tail effect when the wavefronts do not fit one round)
        for (int g : {G * 2, G * 4})
          if (!p.paired && grp_fits(run, sh, g) && g <= std::max(2, 2 * run.maxq)) cand.push_back(g);
        if (cand.size() > 1)
          G = fastest(cand, [&](const KernelArgs &t, int g) { launch_grp(t, run, sh, g, stream); });
        run.tuned_G = G;
      }
    }
    if (run.force_G && G < run.force_G && std::getenv("MISO_GENERAL_LANES") == nullptr) G = run.force_G;   // size bucket (upload)
    if (run.small && p.paired && std::getenv("MISO_GENERAL_LANES") == nullptr && std::getenv("MISO_GENERAL_LANES_BY_CLASS") == nullptr) {   // the small genes' bucket: eight chains per wavefront, score tables in global memory
      GrpShape sh8 = sh;
      sh8.ts = 0;
      if (grp_fits(run, sh8, 8)) { sh = grp_sh[ri] = sh8; G = 8; }
    }
    // a forced (or odd) choice never exceeds the LDS budget: fewer chains per wavefront instead
    while (G < 64 && !grp_fits(run, sh, G)) G = (G < 2) ? 2 : ((G & (G - 1)) ? 64 : G * 2);
    grp_G[ri] = G;
    if (contest) {
      const int nc = flat_nc[ri];
      const int win = (G == 64) ? 0 : fastest({0, G}, [&](const KernelArgs &t, int g) {
        if (g == 0) launch_flat(t, run, nc, flat_nc_max[ri], stream); else launch_grp(t, run, sh, g, stream); });
      run.tuned_flat = win == 0 ? 1 : 0;
      if (win != 0) flat_nc[ri] = 0;
    }
  }

  if (k2_pair && k2_G > 0) {
    const long waves = (static_cast<long>(n_k2 - n_k2w) * p.noChains + 64 / k2_G - 1) / (64 / k2_G);
    k2_pair = waves <= wave_slots || std::getenv("MISO_K2_PAIR") != nullptr;
  }
  // ---- single-end two-isoform events in one launch with two lane widths (sampler_k2_mix, kernels_k2.hip) ----
  // Only when the batch runs as ONE round of paired 8-wavefront workgroups (k2_pair) that does not fill the CUs:
  // the heaviest events get G + 1 lanes per chain, as many of them as still leave one CU per workgroup.
  // MISO_K2_MIX=0 switches it off (A/B), MISO_K2_SPLIT=n forces the number of events of the wide part.
  int k2_mix = 0, k2_mix_blocks = 0;
  {
    const int count = n_k2 - n_k2w, C = p.noChains, G = k2_G;
    const char *off = std::getenv("MISO_K2_MIX");
    if (!p.paired && k2_pair && count > 0 && G >= 1 && G <= 7 && n_kernels <= 1 && !(off && std::atoi(off) == 0)) {
      const int cus = wave_slots / 8, cpwA = 64 / (G + 1), cpwB = 64 / G;
      auto blocks = [&](long chains, int cpw) { return static_cast<int>(((chains + cpw - 1) / cpw + 7) / 8); };
      auto total = [&](int y) { return blocks(static_cast<long>(y) * C, cpwA) + blocks(static_cast<long>(count - y) * C, cpwB); };
      if (total(0) < cus && blocks(static_cast<long>(count) * C, cpwB) * 8 >= wave_slots / 2) {
        int y = 0;
        if (const char *env = std::getenv("MISO_K2_SPLIT")) y = std::max(0, std::min(count - 1, std::atoi(env)));
        else {
          int lo = 0, hi = count - 1;          // total(y) grows with y: the largest y that still fits
          while (lo < hi) { const int mid = (lo + hi + 1) / 2; if (total(mid) <= cus) lo = mid; else hi = mid - 1; }
          y = lo;
        }
        if (y > 0 && total(y) <= cus) { k2_mix = y; k2_mix_blocks = blocks(static_cast<long>(y) * C, cpwA); }
      }
    }
  }
  auto k2_mix_name = [&](int G) { return "sampler_k2_mix<" + std::to_string(G + 1) + ", " + std::to_string(G) + ">"; };
  auto launch_k2_mix = [&](KernelArgs ka, int G, hipStream_t st) {
    const int first = n_k2w, count = n_k2 - n_k2w, C = p.noChains, cpwB = 64 / G;
    ka.slot_event = d_slots + first; ka.n_slots = count;
    ka.pair_waves = 1; ka.mix_slots = k2_mix; ka.mix_blocks = k2_mix_blocks;
    const long wavesB = (static_cast<long>(count - k2_mix) * C + cpwB - 1) / cpwB;
    const unsigned grid = static_cast<unsigned>(k2_mix_blocks + (wavesB + 7) / 8);
    switch (G) {
    case 1: hipLaunchKernelGGL((sampler_k2_mix<2, 1>), dim3(grid), dim3(512), 0, st, ka); break;
    case 2: hipLaunchKernelGGL((sampler_k2_mix<3, 2>), dim3(grid), dim3(512), 0, st, ka); break;
    case 3: hipLaunchKernelGGL((sampler_k2_mix<4, 3>), dim3(grid), dim3(512), 0, st, ka); break;
    case 4: hipLaunchKernelGGL((sampler_k2_mix<5, 4>), dim3(grid), dim3(512), 0, st, ka); break;
    case 5: hipLaunchKernelGGL((sampler_k2_mix<6, 5>), dim3(grid), dim3(512), 0, st, ka); break;
    case 6: hipLaunchKernelGGL((sampler_k2_mix<7, 6>), dim3(grid), dim3(512), 0, st, ka); break;
    default: hipLaunchKernelGGL((sampler_k2_mix<8, 7>), dim3(grid), dim3(512), 0, st, ka); break;
    }
    HIP_OK(hipGetLastError());
  };

  // ---- single-end two-isoform events: a lane width per event (sampler_k2_multi, kernels_k2m.hip) ----
  // Default for single-end batches; MISO_K2_MULTI=0 (A/B, tests) or a forced MISO_LANES_PER_CHAIN: the
  // single-width / two-width launches above.  The plan depends on the list, the chains and this kernel's share of
  // the device only: made once per upload.  MISO_K2_COST="block,step1,step2,step3,step4" overrides the cost model,
  // MISO_K2_TARGET=x forces the bound on a wavefront's step (tests: small batches with many widths).
  // (workgroup-wide chains may use several workgroups, coop.hpp; MISO_NO_COOP=1: their own only)
  const int coop_max = coop_enabled() ? COOP_MAX_N : 1;
  // All cooperative workgroups of ONE launch() -- the gene runs' (wide_setup) and the two-isoform plans' -- come out of one
  // budget of COOP_MAX_WGS (coop.hpp): the gene runs take what upload() asked for, at most 5/8 when two-isoform events are
  // in the batch too; the MODE 2 plan gets two thirds of the rest when both two-isoform kernels run, the other plan the
  // remainder.  (Single-end batches have one cooperative user, the two-isoform plan: all of it.)
  int coop_k2_budget = COOP_MAX_WGS, coop_k2w_budget = COOP_MAX_WGS;
  {
    long want = 0;
    for (int i = 0; i < n_gen; i++) { const int ev_i = h_slots[n_k2 + i]; if (coop_n[ev_i] > 1) want += static_cast<long>(coop_n[ev_i]) * p.noChains; }
    if (std::getenv("MISO_COOP_DRAWS") && p.paired && n_gen > 0) want = COOP_MAX_WGS;   // (tests: the table decides chain by chain)
    const long cap = n_k2 > 0 ? COOP_MAX_WGS * 5 / 8 : COOP_MAX_WGS;
    coop_gen_budget = coop_max > 1 ? static_cast<int>(std::min(want, cap)) : 0;
    const int left = COOP_MAX_WGS - coop_gen_budget;
    coop_k2w_budget = (n_k2w > 0 && n_k2 - n_k2w > 0) ? left * 2 / 3 : left;
    coop_k2_budget = n_k2w > 0 ? left - coop_k2w_budget : left;
    if (!p.paired) coop_k2_budget = left;
  }
  bool k2_multi = false;
  const bool lane_route = collapsed && !p.paired && n_k2 > 0;   // miso_batch_set_collapsed
  // lanes per chain of the collapsed step: one (sampler_lane).  Sharing a chain between 2, 4 or 8 lanes (sampler_k2c:
  // the Metropolis-Hastings step's transcendentals one per lane, the binomial's rejection trials G at a time) brings more
  // wavefronts but the same instructions per wavefront -- 40 000 chains: 56.4 ms on one lane, 58.5 / 59.1 / 86.3 ms on
  // 2 / 4 / 8 (profiles/r03_collapsed.txt); MISO_COLLAPSED_LANES forces (experiments, tests)
  int lane_G = 1;
  if (lane_route) {
    if (const char *f = std::getenv("MISO_COLLAPSED_LANES")) { const int g = std::atoi(f); if (g == 1 || g == 2 || g == 4 || g == 8) lane_G = g; }
  }
  // one chain per lane, up to two wavefronts per SIMD: the form that overlaps a chain's own work (kernels_lane.hip
  // sampler_lane_ilp); beyond, the lean one (six wavefronts per SIMD hide each other's latencies).  MISO_LANE_ILP=0 / 1 forces.
  bool lane_ilp = false;
  if (lane_route && lane_G == 1) {
    lane_ilp = ((static_cast<long>(n_k2) * p.noChains + 255) / 256) * 4 <= wave_slots;
    if (const char *env = std::getenv("MISO_LANE_ILP")) lane_ilp = std::atoi(env) != 0;
  }
  {
    const int count = n_k2 - n_k2w;
    const char *off = std::getenv("MISO_K2_MULTI");
    // paired-end events that keep MODE 1 (a drawing read touches a non-finite score): the same, 4-wavefront workgroups,
    // 4 ... 64 lanes as far as the chains' tables (2 il int32 each) fit the LDS, 256-lane chains
    const long pe1_left = static_cast<long>(LDS_MAX) - static_cast<long>(k2_fp) - K2_RED_BYTES;
    const int pe1_cpw = p.paired ? static_cast<int>(std::max<long>(0, pe1_left / static_cast<long>(4 * std::max<size_t>(k2_tab, 1)))) : 64;
    if (p.paired && count > 0 && pe1_cpw >= 1 && std::getenv("MISO_LANES_PER_CHAIN") == nullptr && !(off && std::atoi(off) == 0)) {
      const int resident = std::max(1, slots_for(static_cast<long>(count) * p.noChains) / 4);
      const long key = (static_cast<long>(resident) * 64 + p.noChains) * 256 + coop_k2_budget + (coop_max > 1 ? 0 : 200);
      if (k2_plan_key != key || std::getenv("MISO_K2_TARGET") || std::getenv("MISO_K2_COST") || std::getenv("MISO_COOP_MIN_QUADS")) {
        static const int widths[] = {4, 8, 16, 32, 64};
        LaneCost cost = k2_cost_paired();
        if (const char *env = std::getenv("MISO_COOP_MIN_QUADS")) cost.coop_min_quads = std::max(1, std::atoi(env));
        std::vector<int> nd(count);
        for (int i = 0; i < count; i++) nd[i] = events[h_slots[n_k2w + i]].n_draw;
        const double forced = std::getenv("MISO_K2_TARGET") ? std::atof(std::getenv("MISO_K2_TARGET")) : 0.0;
        k2_plan = plan_lanes(nd.data(), count, p.noChains, widths, 5, 4, 4, resident, pe1_cpw, cost, forced, coop_max, coop_k2_budget);
        k2_plan_key = key;
      }
      k2_multi = k2_plan.n_segs > 0;
    }
    if (!p.paired && count > 0 && std::getenv("MISO_LANES_PER_CHAIN") == nullptr && !(off && std::atoi(off) == 0)) {
      const int resident = std::max(1, slots_for(static_cast<long>(count) * p.noChains) / 8);
      const long key = (static_cast<long>(resident) * 64 + p.noChains) * 256 + coop_k2_budget + (coop_max > 1 ? 0 : 200);
      if (k2_plan_key != key || std::getenv("MISO_K2_TARGET") || std::getenv("MISO_K2_COST") || std::getenv("MISO_COOP_MIN_QUADS") || std::getenv("MISO_K2_WPB")) {
        static const int widths[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 32, 64};
        LaneCost cost = k2_cost_single();
        if (const char *env = std::getenv("MISO_K2_COST"))
          std::sscanf(env, "%lf,%lf,%lf,%lf,%lf", &cost.block, &cost.step[1], &cost.step[2], &cost.step[3], &cost.step[4]);
        if (const char *env = std::getenv("MISO_COOP_MIN_QUADS")) cost.coop_min_quads = std::max(1, std::atoi(env));
        std::vector<int> nd(count);
        for (int i = 0; i < count; i++) nd[i] = events[h_slots[n_k2w + i]].n_draw;
        const double forced = std::getenv("MISO_K2_TARGET") ? std::atof(std::getenv("MISO_K2_TARGET")) : 0.0;
        k2_plan = plan_lanes(nd.data(), count, p.noChains, widths, 12, 8, 8, resident, 64, cost, forced, coop_max, coop_k2_budget);
        // several rounds: smaller workgroups (a workgroup starts when ALL its wavefronts' slots are free; with 8 the
        // slots of early finishers idle: MISO defaults on 40 000 events 425 ms, with 4: see profiles/r03_k2_multi_ab.txt)
        int wpb = k2_plan.rounds == 1 ? 8 : 4;
        if (const char *env = std::getenv("MISO_K2_WPB")) wpb = std::atoi(env);
        if (wpb != 8 && wpb != 4 && wpb != 1) MISO_FAIL(MISO_EINVAL, "MISO_K2_WPB must be 8, 4 or 1");
        if (wpb != 8) k2_plan = plan_lanes(nd.data(), count, p.noChains, widths, 12, wpb >= 4 ? wpb : 0, wpb, resident * 8 / wpb, 64, cost, forced, coop_max, coop_k2_budget);
        k2_plan_key = key;
        // One round, 8 wavefronts per workgroup: wavefronts w and w + 4 of a workgroup share a SIMD.  Pair the launch's
        // wavefronts (all runs but the workgroup-wide chains') by estimated duration, the i-th heaviest with the i-th
        // lightest ACROSS the runs: within a run the p-th heaviest + p-th lightest (a.pair_waves) give every SIMD of the
        // run the same work, but the runs' sums differ (a run's first wavefront sits at the planner's bound, its last
        // wherever the next width takes over) and the SIMDs of the lighter runs idle at the end (wave-slot occupancy
        // 0.78, profiles/r03_wave_time.txt).  Measured, 40 000 events: hg19-like read counts (13 runs) 73.9 -> 71.4 ms;
        // 1000 reads each (4 runs) 99.7 -> 100.8 ms, 3000 reads 244.6 -> 250.0 ms (few runs: their sums are close already,
        // and the kernel with a body per run AND per wavefront is larger): used from seven runs on.
        // MISO_K2_GLOBAL_PAIR=0 / 1: never / always (A/B, tests).
        k2_pair_tab.clear(); k2_pair_grid = 0; k2_pair_wide_blocks = 0;
        const char *gp = std::getenv("MISO_K2_GLOBAL_PAIR");
        if (k2_plan.n_segs > 0 && k2_plan.rounds == 1 && wpb == 8 && (gp ? std::atoi(gp) != 0 : k2_plan.n_segs >= 7)) {
          struct W { double cost; int32_t code; };
          std::vector<W> ws;
          bool ok = true, seen_narrow = false;
          for (int sg = 0; sg < k2_plan.n_segs && ok; sg++) {
            const int G = k2_plan.seg_lanes[sg];
            if (G == K2_WIDE) { ok = !seen_narrow; k2_pair_wide_blocks = k2_plan.seg_block[sg + 1]; continue; }   // (wide runs come first)
            seen_narrow = true;
            const int cpw = 64 / G;
            const long chains = static_cast<long>(k2_plan.seg_slot[sg + 1] - k2_plan.seg_slot[sg]) * p.noChains;
            const long waves = (chains + cpw - 1) / cpw;
            ok = waves < (1L << 20) && sg < 2048;
            for (long w = 0; w < waves && ok; w++) {
              const int first = nd[k2_plan.seg_slot[sg] + static_cast<int>((w * cpw) / p.noChains)];   // the wavefront's longest chain
              ws.push_back(W{cost.wave_step(G, first), static_cast<int32_t>(sg << 20 | static_cast<int32_t>(w))});
            }
          }
          if (ok && !ws.empty()) {
            std::stable_sort(ws.begin(), ws.end(), [](const W &x, const W &y) { return x.cost > y.cost; });
            const size_t N = ws.size(), Q = (N + 1) / 2, blocks = (Q + 3) / 4;
            k2_pair_tab.assign(blocks * 8, -1);
            for (size_t q = 0; q < Q; q++) {
              k2_pair_tab[(q / 4) * 8 + q % 4] = ws[q].code;
              if (N - 1 - q > q) k2_pair_tab[(q / 4) * 8 + q % 4 + 4] = ws[N - 1 - q].code;
            }
            k2_pair_grid = k2_pair_wide_blocks + static_cast<int>(blocks);
            if (d_k2_pair_tab) HIP_OK(hipFree(d_k2_pair_tab));
            HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_k2_pair_tab), k2_pair_tab.size() * sizeof(int32_t)));
            HIP_OK(hipMemcpy(d_k2_pair_tab, k2_pair_tab.data(), k2_pair_tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
          }
        }
      }
      k2_multi = k2_plan.n_segs > 0;
    }
  }
  // ---- paired-end two-isoform events of MODE 2, likewise (sampler_k2_multi<2, 4>): 4, 8, ... 64 lanes per chain as far
  // as the chains' score tables fit the workgroup's LDS, 256 lanes (the workgroup) for the largest events ----
  bool k2w_multi = false;
  size_t k2w_multi_lds = 0;
  {
    const char *off = std::getenv("MISO_K2_MULTI");
    if (p.paired && n_k2w > 0 && std::getenv("MISO_LANES_PER_CHAIN") == nullptr && !(off && std::atoi(off) == 0)) {
      const long lds_left = static_cast<long>(LDS_MAX) - static_cast<long>(k2w_fp) - K2_RED_BYTES;
      const int max_cpw = static_cast<int>(std::max<long>(0, lds_left / static_cast<long>(4 * k2w_tab)));
      const int resident = std::max(1, slots_for(static_cast<long>(n_k2w) * p.noChains) / 4);
      const long key = (static_cast<long>(resident) * 64 + p.noChains) * 256 + coop_k2w_budget + (coop_max > 1 ? 0 : 200);
      if (max_cpw >= 1 && (k2w_plan_key != key || std::getenv("MISO_K2_TARGET") || std::getenv("MISO_K2_COST") || std::getenv("MISO_COOP_MIN_QUADS") || std::getenv("MISO_K2W_WPB"))) {
        static const int widths[] = {4, 8, 16, 32, 64};
        LaneCost cost = k2_cost_paired();
        if (const char *env = std::getenv("MISO_K2_COST"))
          std::sscanf(env, "%lf,%lf,%lf,%lf,%lf", &cost.block, &cost.step[1], &cost.step[2], &cost.step[3], &cost.step[4]);
        if (const char *env = std::getenv("MISO_COOP_MIN_QUADS")) cost.coop_min_quads = std::max(1, std::atoi(env));
        std::vector<int> nd(n_k2w);
        for (int i = 0; i < n_k2w; i++) nd[i] = events[h_slots[i]].n_draw;
        const double forced = std::getenv("MISO_K2_TARGET") ? std::atof(std::getenv("MISO_K2_TARGET")) : 0.0;
        k2w_plan = plan_lanes(nd.data(), n_k2w, p.noChains, widths, 5, 4, 4, resident, max_cpw, cost, forced, coop_max, coop_k2w_budget);
        // The widest a chain can be is its workgroup: 256 lanes with 4 wavefronts, 512 with 8 (one workgroup per CU,
        // twice the LDS each: the same chains per CU).  8-wavefront workgroups lose ~19 % when the launch runs in
        // several rounds (a workgroup starts when ALL its slots are free), so they are taken only when the largest
        // events would otherwise outlast the launch (hg19-like read counts: 218 ms -> 175 ms; uniform: 217 ms -> 259 ms, not taken; profiles/r03_pe_k2_wpb.txt).
        {
          const LanePlan p8 = plan_lanes(nd.data(), n_k2w, p.noChains, widths, 5, 8, 8, std::max(1, resident / 2), max_cpw, cost, forced, coop_max, coop_k2w_budget);
          const char *env = std::getenv("MISO_K2W_WPB");
          if (env ? std::atoi(env) == 8 : 1.2 * p8.est < k2w_plan.est) k2w_plan = p8;
        }
        k2w_plan_key = key;
      }
      k2w_multi = max_cpw >= 1 && k2w_plan.n_segs > 0;
      if (k2w_multi) {
        int cpw = 1;
        for (int i = 0; i < k2w_plan.n_segs; i++)
          if (k2w_plan.seg_lanes[i] != K2_WIDE) cpw = std::max(cpw, 64 / k2w_plan.seg_lanes[i]);
        k2w_multi_lds = align_up(k2w_fp + static_cast<size_t>(k2w_plan.wpb) * cpw * k2w_tab, 16);
      }
    }
  }
  // the several-rounds single-end plan has runs of one and two lanes per chain only: the kernel of three wavefronts per SIMD (kernels_k2m.inl)
  bool k2_narrow = k2_multi && !p.paired && k2_plan.wpb == 4 && k2_plan.n_segs > 0 && std::getenv("MISO_K2_NO_NARROW") == nullptr;
  for (int i = 0; i < k2_plan.n_segs && k2_narrow; i++) k2_narrow = k2_plan.seg_lanes[i] <= 2;
  const std::string k2m_name = "sampler_k2_multi<" + std::string(p.paired ? "1, " : "0, ") + std::to_string(k2_plan.wpb) + (k2_narrow ? ", true>" : ">");
  // which chain every workgroup of a plan's workgroup-wide run works on (coop.hpp); nothing to do when every chain
  // has its workgroup to itself
  auto k2_coop = [&](KernelArgs &ka, const LanePlan &pl, K2Coop &cc, hipStream_t st) {
    ka.coop_tab = nullptr; ka.coop_mem = nullptr;
    if (pl.n_segs == 0 || pl.seg_lanes[0] != K2_WIDE) return;
    bool any = false;
    for (int m : pl.wide_wgs) any |= m > 1;
    if (!any) return;
    if (cc.key != pl.seg_block[1] * 131 + static_cast<long>(pl.wide_wgs.size()) || !cc.d_tab) {
      std::vector<int32_t> tab;
      cc.chains = 0;
      for (size_t e = 0; e < pl.wide_wgs.size(); e++)
        for (int c = 0; c < p.noChains; c++) {
          const int m = pl.wide_wgs[e];
          for (int r = 0; r < m; r++) { tab.push_back(static_cast<int32_t>(e * p.noChains + c)); tab.push_back(r); tab.push_back(m); tab.push_back(m > 1 ? cc.chains : 0); }
          if (m > 1) cc.chains++;
        }
      if (static_cast<int>(tab.size() / 4) != pl.seg_block[1]) MISO_FAIL(MISO_EINTERNAL, "lane plan and cooperative table disagree");
      if (cc.d_tab) (void) hipFree(cc.d_tab);
      if (cc.d_mem) (void) hipFree(cc.d_mem);
      cc.d_tab = nullptr; cc.d_mem = nullptr;
      HIP_OK(hipMalloc(reinterpret_cast<void **>(&cc.d_tab), tab.size() * sizeof(int32_t)));
      HIP_OK(hipMemcpy(cc.d_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      HIP_OK(hipMalloc(reinterpret_cast<void **>(&cc.d_mem), std::max(1, cc.chains) * COOP_WORDS * sizeof(uint32_t)));
      cc.key = pl.seg_block[1] * 131 + static_cast<long>(pl.wide_wgs.size());
    }
    HIP_OK(hipMemsetAsync(cc.d_mem, 0, std::max(1, cc.chains) * COOP_WORDS * sizeof(uint32_t), st));
    ka.coop_tab = cc.d_tab; ka.coop_mem = cc.d_mem;
  };
  auto launch_k2_multi = [&](KernelArgs ka, hipStream_t st, bool wpart = false) {
    k2_coop(ka, wpart ? k2w_plan : k2_plan, wpart ? k2w_coop : k2_coop_se, st);
    if (wpart) {
      ka.slot_event = d_slots; ka.n_slots = n_k2w;
      ka.pair_waves = 0;
      ka.red_off = static_cast<int32_t>(k2w_multi_lds);
      ka.n_segs = k2w_plan.n_segs;
      for (int i = 0; i <= k2w_plan.n_segs; i++) { ka.seg_block[i] = k2w_plan.seg_block[i]; ka.seg_slot[i] = k2w_plan.seg_slot[i]; }
      for (int i = 0; i < k2w_plan.n_segs; i++) ka.seg_lanes[i] = k2w_plan.seg_lanes[i];
      const int lds = static_cast<int>(k2w_multi_lds + K2_RED_BYTES);
      const dim3 grid(static_cast<unsigned>(k2w_plan.seg_block[k2w_plan.n_segs]));
      if (k2w_plan.wpb == 8) {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_k2_multi<2, 8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((sampler_k2_multi<2, 8>), grid, dim3(512), lds, st, ka);
      } else {
        HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_k2_multi<2, 4>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL((sampler_k2_multi<2, 4>), grid, dim3(256), lds, st, ka);
      }
      HIP_OK(hipGetLastError());
      return;
    }
    ka.slot_event = d_slots + n_k2w; ka.n_slots = n_k2 - n_k2w;
    ka.pair_waves = (k2_plan.rounds == 1 && k2_plan.wpb == 8) ? 1 : 0;
    ka.red_off = 0;
    ka.n_segs = k2_plan.n_segs;
    for (int i = 0; i <= k2_plan.n_segs; i++) { ka.seg_block[i] = k2_plan.seg_block[i]; ka.seg_slot[i] = k2_plan.seg_slot[i]; }
    for (int i = 0; i < k2_plan.n_segs; i++) ka.seg_lanes[i] = k2_plan.seg_lanes[i];
    dim3 grid(static_cast<unsigned>(k2_plan.seg_block[k2_plan.n_segs]));
    if (!p.paired && k2_plan.wpb == 8 && k2_pair_grid > 0 && !k2_pair_tab.empty()) {   // wavefronts paired across the runs
      ka.wave_tab = d_k2_pair_tab; ka.mix_blocks = k2_pair_wide_blocks; ka.pair_waves = 0;
      grid = dim3(static_cast<unsigned>(k2_pair_grid));
    }
    if (p.paired) {   // MODE 1
      int cpw = 1;
      for (int i = 0; i < k2_plan.n_segs; i++) if (k2_plan.seg_lanes[i] != K2_WIDE) cpw = std::max(cpw, 64 / k2_plan.seg_lanes[i]);
      const size_t tabs = align_up(k2_fp + 4 * static_cast<size_t>(cpw) * k2_tab, 16);
      ka.pair_waves = 0;
      ka.red_off = static_cast<int32_t>(tabs);
      const int lds = static_cast<int>(tabs + K2_RED_BYTES);
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_k2_multi<1, 4>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL((sampler_k2_multi<1, 4>), grid, dim3(256), lds, st, ka);
      HIP_OK(hipGetLastError());
      return;
    }
    switch (k2_plan.wpb) {
    case 8:
      // the whole launch resident at once: the two wavefronts of a SIMD keep step (kernels_k2.inl k2_balance)
      if (k2_plan.rounds == 1) ka.balance = (std::getenv("MISO_K2_BALANCE") && std::atoi(std::getenv("MISO_K2_BALANCE")) == 0) ? 0 : 1;
      hipLaunchKernelGGL((sampler_k2_multi<0, 8>), grid, dim3(512), K2_RED_BYTES, st, ka);
      break;
    case 4:
      if (k2_narrow) hipLaunchKernelGGL((sampler_k2_multi<0, 4, true>), grid, dim3(256), K2_RED_BYTES, st, ka);
      else hipLaunchKernelGGL((sampler_k2_multi<0, 4>), grid, dim3(256), K2_RED_BYTES, st, ka);
      break;
    default: hipLaunchKernelGGL((sampler_k2_multi<0, 1>), grid, dim3(64), 0, st, ka); break;
    }
    HIP_OK(hipGetLastError());
  };

  auto k2_name = [&](int G, bool wpart) {
    return "sampler_k2<" + std::to_string(G) + (p.paired ? (wpart ? ", 2, 4>" : ", 1, 4>") : (k2_pair ? ", 0, 8>" : ", 0, 4>"));
  };
  // ---- what goes on the device, kernel by kernel (miso_batch_launch_stats): built on demand ----
  const LanePlan plan_copy = k2_plan, planw_copy = k2w_plan;
  stats_builder = [=]() {
  kernel_stats.clear();
  auto add_stat = [&](const std::string &name, double waves, double trips, double chains, double words) {
    miso_kernel_stat_t ks{};
    std::snprintf(ks.name, sizeof ks.name, "%s", name.c_str());
    ks.waves = waves; ks.trips = trips; ks.iterations = static_cast<double>(p.noIterations) + 1.0;
    ks.chains = chains; ks.words = words;
    kernel_stats.push_back(ks);
  };
  auto k2_name = [&](int G, bool wpart) {
    return "sampler_k2<" + std::to_string(G) + (p.paired ? (wpart ? ", 2, 4>" : ", 1, 4>") : (k2_pair ? ", 0, 8>" : ", 0, 4>"));
  };
  auto k2_mix_name = [&](int G) { return "sampler_k2_mix<" + std::to_string(G + 1) + ", " + std::to_string(G) + ">"; };
  // up to twenty isoforms the kernel of the launch's largest isoform count (its slice layout at compile time,
  // kernels_flat.inl); MISO_FLAT_NO_KS=1: the run-time layout everywhere (A/B, tests)
  auto flat_ks = [&](const GenRun &run) {
    const int lo = run.kc == 4 ? 3 : (run.kc == 8 ? 5 : (run.kc == 12 ? 9 : (run.kc == 16 ? 13 : 17)));
    return (run.kc <= 32 && run.kmax >= lo && run.kmax <= std::min(run.kc, 20) && std::getenv("MISO_FLAT_NO_KS") == nullptr) ? run.kmax : 0;
  };
  // ... and, when every event of the launch has that count, the kernel that knows it (UNI: no per-chain `k < K` masks)
  auto flat_uni = [&](const GenRun &run) { return flat_ks(run) > 0 && run.kmin == run.kmax && std::getenv("MISO_FLAT_NO_UNI") == nullptr; };
  auto flat_name = [&](const GenRun &run) {
    return "sampler_flat<" + std::to_string(run.kc) + ", " + std::to_string(flat_ks(run)) + (flat_uni(run) ? ", true>" : ">");
  };
  for (int part = 0; part < 2; part++) {
    const bool wpart = part == 0;
    const int count = wpart ? n_k2w : n_k2 - n_k2w, k2G = wpart ? k2w_G : k2_G;
    if (count <= 0) continue;
    // slot order = events by drawing reads, descending, each with its noChains chains
    std::vector<int> nd;
    for (const PackedEvent &e : events) if (e.K == 2 && !k2_general && (use_delta && e.pe_delta && !e.draw_dense.empty()) == wpart) nd.push_back(e.n_draw);
    std::sort(nd.begin(), nd.end(), [](int x, int y) { return x > y; });
    const int C = p.noChains;
    const long chains = static_cast<long>(count) * C;
    double trips = 0, words = 0;
    long waves = 0;
    const int bshift = p.paired ? 2 : 3;   // draws per Philox block: four, single-end eight (half-words, miso_philox.h)
    auto slice = [&](long c0, long c1, int G) {   // chains [c0, c1) of the list with G lanes per chain
      const int cpw = 64 / G;
      for (long s0 = c0; s0 < c1; s0 += cpw, waves++) {
        int mx = 0, any_rem = 0;
        for (long sl = s0; sl < std::min(c1, s0 + cpw); sl++) {
          const int n = nd[sl / C];
          mx = std::max(mx, n >> bshift); any_rem |= n & ((1 << bshift) - 1);
          words += n;
        }
        const int t = (mx + 2 * G - 1) / (2 * G);             // trips of two Philox blocks per lane
        trips += p.paired ? 2 * t + 1 : 2 * t + (any_rem ? 1 : 0);   // counted in blocks per lane
      }
    };
    if (!wpart && lane_route) {
      for (int n : nd) words += static_cast<double>(n) * C;
      add_stat(lane_G == 1 ? std::string(lane_ilp ? "sampler_lane_ilp" : "sampler_lane") : "sampler_k2c<" + std::to_string(lane_G) + ">",
               static_cast<double>((chains * lane_G + 63) / 64), 0.0, static_cast<double>(chains), words);
    } else if (wpart ? k2w_multi : k2_multi) {
      const LanePlan &pl = wpart ? planw_copy : plan_copy;
      for (int sg = 0; sg < pl.n_segs; sg++) {
        const long c0 = static_cast<long>(pl.seg_slot[sg]) * C, c1 = static_cast<long>(pl.seg_slot[sg + 1]) * C;
        if (pl.seg_lanes[sg] == K2_WIDE) {
          for (long c = c0; c < c1; c++) {
            const int n = nd[c / C];
            const size_t we = static_cast<size_t>(c / C - pl.seg_slot[sg]);
            const int m = we < pl.wide_wgs.size() ? pl.wide_wgs[we] : 1;     // workgroups of the chain (coop.hpp)
            const int wl = 64 * pl.wpb * m;
            const int t = ((n >> bshift) + 2 * wl - 1) / (2 * wl);
            trips += pl.wpb * m * (p.paired ? 2 * t + 1 : 2 * t + ((n & 7) ? 1 : 0));
            words += n; waves += pl.wpb * m;
          }
        } else slice(c0, c1, pl.seg_lanes[sg]);
      }
      add_stat(wpart ? "sampler_k2_multi<2, " + std::to_string(pl.wpb) + ">" : k2m_name, static_cast<double>(waves), trips,
               static_cast<double>(chains), words);
    } else if (!wpart && k2_mix > 0) {
      slice(0, static_cast<long>(k2_mix) * C, k2G + 1);
      slice(static_cast<long>(k2_mix) * C, chains, k2G);
      add_stat(k2_mix_name(k2G), static_cast<double>(waves), trips, static_cast<double>(chains), words);
    } else {
      slice(0, chains, k2G);
      add_stat(k2_name(k2G, wpart), static_cast<double>(waves), trips, static_cast<double>(chains), words);
    }
  }
  for (size_t ri = 0; ri < gen_runs.size(); ri++) {
    const GenRun &run = gen_runs[ri];
    if (run_lane[ri]) {
      double words = 0;
      for (int i = 0; i < run.count; i++) words += static_cast<double>(events[h_slots[n_k2 + run.first + i]].n_draw) * p.noChains;
      const long chains = static_cast<long>(run.count) * p.noChains;
      add_stat("sampler_lane_k", static_cast<double>((chains + 63) / 64), 0.0, static_cast<double>(chains), words);
      continue;
    }
    const bool flat = flat_nc[ri] > 0;
    const int G = grp_G[ri], C = p.noChains, cpw = flat ? flat_nc[ri] : std::max(1, 64 / G);
    const bool w64 = G == 64 && run.wave64 && p.paired && run.dense;
    const long chains = static_cast<long>(run.count) * C;
    std::vector<const PackedEvent *> evs;   // the run's events in slot order
    for (int j = 0; j < run.count; j++) evs.push_back(&events[h_slots[n_k2 + run.first + j]]);   // the list upload made
    const bool cls = !p.paired && grp_sh[ri].qs > 0;
    double trips = 0, words = 0;
    long waves = 0;
    if (flat) {   // the wavefronts as flat_waves laid them out
      for (size_t w = 0; w + 1 < run.wave_tab.size(); w += 2) {
        const int first = run.wave_tab[w], n = run.wave_tab[w + 1] & 0xFF;
        const bool wide = (run.wave_tab[w + 1] & FLAT_WIDE) != 0;
        if (n == 0) continue;
        long units = 0;
        for (long sl = first; sl < first + n; sl++) { units += evs[sl / C]->n_units; if (!wide || (w / 2) % 4 == 0) words += evs[sl / C]->n_draw; }
        trips += wide ? (units + 255) / 256 : (units + 63) / 64;
        waves++;
      }
    } else
    for (long s0 = 0; s0 < chains; s0 += cpw, waves++) {
      int mx = 0, kmx = 0; long units = 0;
      for (long sl = s0; sl < std::min(chains, s0 + cpw); sl++) {
        const PackedEvent &e = *evs[sl / C];
        mx = std::max(mx, cls ? e.n_units : (e.n_draw + 3) / 4);
        units += e.n_units;
        kmx = std::max(kmx, e.K);
        words += e.n_draw;
      }
      const int nw = (cls && kmx - 1 <= 3) ? 2 : 1;         // Philox blocks in flight (class path, K <= 4)
      trips += flat ? (units + 63) / 64 : ((G == 64 && !w64) ? mx : nw * ((mx + nw * G - 1) / (nw * G)));
    }
    if (run.wide) {   // four wavefronts per chain, the quads dealt over 256 lanes
      trips = 0; waves = 0;
      for (long sl = 0; sl < chains; sl++) { trips += 4 * ((((evs[sl / C]->n_draw + 3) / 4) + 255) / 256); waves += 4; }
    }
    const std::string name = (ri < run_in_multi.size() && run_in_multi[ri] == 2) ? std::string("sampler_grp_all") :
                             (ri < run_in_multi.size() && run_in_multi[ri]) ? "sampler_grp_multi<" + std::to_string(run.kc) + ">" :
                             run.wide ? "sampler_grp<64, true, " + std::to_string(run.kc) + ", true>" :
                             flat ? flat_name(run) : ((G == 64 && !w64) ? std::string(run.kc > 64 ? "sampler_big<" : "sampler_wave<") : "sampler_grp<" + std::to_string(G) + ", ") +
                             (p.paired ? "true" : "false") + ((G == 64 && !w64) ? std::string(">") : ", " + std::to_string(run.kc) + ">");
    add_stat(name, static_cast<double>(waves), trips, static_cast<double>(chains), words);
  }
  };   // stats_builder

  HIP_OK(hipEventRecord(ev0, stream));                  // events bracket the sampler kernels only
  // kernel i > 0 goes to its own stream, forked from and joined back into the batch's stream
  size_t kernel_no = 0;
  auto stream_for_next = [&]() {
    const size_t i = kernel_no++;
    if (i == 0 || std::getenv("MISO_SERIAL_KERNELS") != nullptr) return stream;
    while (aux_streams.size() < i) {
      hipStream_t st; hipEvent_t e;
      // (numerically lower = higher priority; the range has three levels on this device: the second and third kernel
      // of a launch at the middle one, the rest at the lowest)
      const int n_aux = static_cast<int>(aux_streams.size());
      const int pr = prio_hi == prio_lo ? prio_lo : std::min(prio_lo, prio_hi + 1 + n_aux / 2);
      if (prio_hi != prio_lo) HIP_OK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr));
      else HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      aux_streams.push_back(st); aux_done.push_back(e);
    }
    HIP_OK(hipStreamWaitEvent(aux_streams[i - 1], ev0, 0));
    return aux_streams[i - 1];
  };
  // chains on several workgroups first: their workgroups must all be resident at once (coop.hpp), which the idle
  // device guarantees; what is launched after them never waits for them
  // A size bucket whose lanes came out the same as the class's normal launch joins it: one launch (its chains stay
  // in front: longest first), one hardware queue less to wait in (kernels of different streams share the process's
  // hardware queues, and a kernel waits for the one before it in its queue).
  std::vector<char> merged_into_prev(gen_runs.size(), 0);
  // (measured, 16 384 genes of 3-20 isoforms: 4 queues 290 -> 254 ms merged; 8 queues 195 ms apart, 241 ms merged --
  // the separate launches stagger the classes' starts; capi.hip asks for 8 queues)
  // (what is IN EFFECT, not what the environment says now: capi.hip's constructor records it; 0 = unknown -- the runtime was
  // up before this library was loaded -- counts as the runtime's default of 4)
  const int hwq = g_hw_queues_in_effect > 0 ? g_hw_queues_in_effect : 4;
  const bool merge_on = std::getenv("MISO_NO_PE_MERGE") == nullptr && (std::getenv("MISO_PE_MERGE") != nullptr || hwq < 8);
  auto launch_gen_run = [&](size_t ri) {
    GenRun &run = gen_runs[ri];
    const int G = grp_G[ri];
    if (merged_into_prev[ri]) return;
    const bool wave_k = G == 64 && !(run.wave64 && p.paired && run.dense);
    if (merge_on && !run.wide && run.force_G && G != 64 && flat_nc[ri] == 0 && ri + 1 < gen_runs.size()) {
      const GenRun &nx = gen_runs[ri + 1];
      if (!nx.wide && !nx.force_G && nx.kc == run.kc && grp_G[ri + 1] == G && flat_nc[ri + 1] == 0 &&
          nx.first == run.first + run.count) {
        GenRun m;
        m.first = run.first; m.count = run.count + nx.count; m.kc = run.kc;
        m.kmax = std::max(run.kmax, nx.kmax); m.kmin = std::min(run.kmin, nx.kmin); m.maxq = std::max(run.maxq, nx.maxq);
        m.maxcls = std::max(run.maxcls, nx.maxcls); m.nocls = run.nocls || nx.nocls; m.dense = run.dense && nx.dense;
        const GrpShape msh = grp_shape(m);
        if (grp_fits(m, msh, G)) {
          merged_into_prev[ri + 1] = 1;
          last_kernels += std::string(last_kernels.empty() ? "" : ",") + "sampler_grp<" + std::to_string(G) + ", " +
                          (p.paired ? "true" : "false") + ", " + std::to_string(run.kc) + ">";
          launch_grp(a, m, msh, G, stream_for_next());
          return;
        }
      }
    }
    if (flat_nc[ri] > 0) {
      last_kernels += std::string(last_kernels.empty() ? "" : ",") + flat_name(run);
      launch_flat(a, run, flat_nc[ri], flat_nc_max[ri], stream_for_next());
      return;
    }
    last_kernels += std::string(last_kernels.empty() ? "" : ",") +
                    (run.wide ? "sampler_grp<64, true, " + std::to_string(run.kc) + ", true>"
                              : (wave_k ? std::string(run.kc > 64 ? "sampler_big<" : "sampler_wave<")
                                        : "sampler_grp<" + std::to_string(G) + ", ") +
                                    (p.paired ? "true" : "false") +
                                    (wave_k ? std::string(">") : ", " + std::to_string(run.kc) + ">"));
    launch_grp(a, run, grp_sh[ri], G, stream_for_next());
  };
  auto ensure_logfact = [&](int first, int count) {   // log factorials (miso_binomial.h) up to the largest event's drawing reads
    int need = 2;
    for (int i = 0; i < count; i++) need = std::max(need, events[h_slots[first + i]].n_draw + 2);
    if (need > logfact_n) {
      std::vector<double> lf(static_cast<size_t>(need));
      miso_logfact_fill(lf.data(), need);
      HIP_OK(hipStreamSynchronize(stream));   // (an earlier launch of this batch may still read the old table)
      if (d_logfact) HIP_OK(hipFree(d_logfact));
      HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_logfact), lf.size() * sizeof(double)));
      HIP_OK(hipMemcpy(d_logfact, lf.data(), lf.size() * sizeof(double), hipMemcpyHostToDevice));
      logfact_n = need;
    }
  };
  if (lane_route || lane_gen) ensure_logfact(0, n_k2 + n_gen);
  for (size_t ri = 0; ri < gen_runs.size(); ri++) {
    if (!run_lane[ri]) continue;
    const GenRun &run = gen_runs[ri];
    const int ks = std::max(2, run.kmax);
    KernelArgs ka = a;
    ka.slot_event = d_slots + n_k2 + run.first; ka.n_slots = run.count; ka.kstride = ks; ka.logfact = d_logfact;
    const long chains = static_cast<long>(run.count) * p.noChains;
    const size_t lds = lanek_lds_bytes(ks);
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_lane_k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    last_kernels += std::string(last_kernels.empty() ? "" : ",") + "sampler_lane_k";
    hipLaunchKernelGGL(sampler_lane_k, dim3(static_cast<unsigned>((chains + 63) / 64)), dim3(64), lds, stream_for_next(), ka);
    HIP_OK(hipGetLastError());
  }
  // Paired-end: the size buckets of one isoform-count class in ONE launch (sampler_grp_multi<KC>, kernels_grp.inl):
  // segments = the class's runs in list order -- workgroup-wide chains (those on several workgroups in front), a
  // wavefront per chain, 32 lanes, 16 lanes.  MISO_NO_PE_MULTI=1: every run its own launch (A/B, tests).
  std::vector<char> in_multi(gen_runs.size(), 0);
  bool all_launched = false;
  // Paired-end, two classes or more, every run on dense records with sixteen lanes per chain (a whole-gene batch of like-sized
  // genes): ALL of them in one launch (sampler_grp_all, kernels_grp.inl), the segments ordered by what a workgroup costs --
  // (quads per lane + the scalar step) x isoforms, descending: longest first across the classes.  Every segment keeps the run's
  // own slice layout (launch_grp's).  Batches with size buckets (real pair counts) keep a launch per class
  // (sampler_grp_multi): twenty bodies in one kernel do not compile in reasonable time (round 5: 28 minutes, 802 spilled registers).
  // MISO_NO_PE_ALL=1: the launches per run below (A/B, tests).
  {
    bool all_ok = p.paired && !lane_gen && gen_runs.size() >= 2 && gen_runs.size() < 127 && std::getenv("MISO_NO_PE_ALL") == nullptr &&
                  std::getenv("MISO_NO_PE_MULTI") == nullptr && std::getenv("MISO_SERIAL_KERNELS") == nullptr &&
                  (std::getenv("MISO_GENERAL_LANES") == nullptr || std::getenv("MISO_PE_ALL") != nullptr);   // (MISO_PE_ALL=1 with MISO_GENERAL_LANES=16: tests on small batches)
    bool two_classes = false;
    for (size_t ri = 0; ri < gen_runs.size() && all_ok; ri++) {
      const GenRun &run = gen_runs[ri];
      const int G = grp_G[ri];
      all_ok = !run_lane[ri] && flat_nc[ri] == 0 && fp_rows(run) && run.kc <= 32 && !run.wide && G == 16;
      two_classes |= run.kc != gen_runs[0].kc;
    }
    if (all_ok && two_classes) {
      struct Seg { size_t ri; int e0, e1; double cost; };
      std::vector<Seg> segs;
      // (a segment per isoform count, so that no wavefront carries two -- see sampler_grp_multi below; MISO_PE_ALL_ORDER=0: a segment per
      // run, 2: heaviest / lightest alternating, 3: lightest first (A/B: 777 / 705 / 767 / 851 ms for 0 / 1 / 2 / 3))
      const int order = std::getenv("MISO_PE_ALL_ORDER") ? std::atoi(std::getenv("MISO_PE_ALL_ORDER")) : 1;
      for (size_t ri = 0; ri < gen_runs.size(); ri++) {
        const GenRun &run = gen_runs[ri];
        if (order == 0) { segs.push_back(Seg{ri, 0, run.count, (static_cast<double>(run.maxq) / 16.0 + 8.0) * (run.kmax + 2)}); continue; }
        auto cost_at = [&](int e) {
          const PackedEvent &ev = events[h_slots[n_k2 + run.first + e]];
          return (static_cast<double>((ev.n_draw + 3) / 4) / 16.0 + 8.0) * (ev.K + 2);
        };
        int e0 = 0;   // pieces: one per isoform count
        for (int e = 1; e <= run.count; e++)
          if (e == run.count || events[h_slots[n_k2 + run.first + e]].K != events[h_slots[n_k2 + run.first + e0]].K) { segs.push_back(Seg{ri, e0, e, cost_at(e0)}); e0 = e; }
      }
      std::stable_sort(segs.begin(), segs.end(), [&](const Seg &x, const Seg &y) { return x.cost > y.cost; });
      if (order == 2) {   // heaviest, lightest, second heaviest, second lightest, ...
        std::vector<Seg> t;
        for (size_t i = 0, j = segs.size(); i < j;) { t.push_back(segs[i++]); if (i < j) t.push_back(segs[--j]); }
        segs = t;
      }
      if (order == 3) std::reverse(segs.begin(), segs.end());   // lightest first
      KernelArgs ka = a;
      hipStream_t st = stream_for_next();
      ka.slot_event = d_slots + n_k2; ka.n_slots = n_gen;
      ka.cstride = 0; ka.pe_dense = 1; ka.pe_force_exact = std::getenv("MISO_PE_FORCE_EXACT") != nullptr;
      size_t lds = 0; int blocks = 0;
      std::vector<GrpSeg> tab;
      for (const Seg &sg : segs) {
        const GenRun &run = gen_runs[sg.ri];
        const GrpShape &sh = grp_sh[sg.ri];
        const long chains = static_cast<long>(sg.e1 - sg.e0) * p.noChains;
        GrpSeg g{};
        g.block0 = blocks; g.slot0 = run.first + sg.e0; g.n_slots = sg.e1 - sg.e0; g.kc = run.kc;
        g.kstride = run.kmax; g.tstride = sh.ts;
        lds = std::max(lds, fp_bytes_of(run) + 4 * 4 * static_cast<size_t>(grp_slice_bytes(run.kmax, 0, sh.ts)));
        blocks += static_cast<int>(((chains + 3) / 4 + 3) / 4);
        tab.push_back(g);
        in_multi[sg.ri] = 2;
      }
      GrpSeg sentinel{}; sentinel.block0 = blocks;
      tab.push_back(sentinel);
      if (lds > LDS_MAX) MISO_FAIL(MISO_EINTERNAL, "sampler_grp_all: a segment's slices exceed the workgroup's LDS");
      // (uploaded when it changed: the batch's first launch, or another upload's events; the launch that read the old one has
      // been waited for by then)
      constexpr size_t GRP_SEGS_CAP = 256;
      if (tab.size() > GRP_SEGS_CAP) MISO_FAIL(MISO_EINTERNAL, "sampler_grp_all: too many segments");
      if (!d_grp_segs) HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_grp_segs), GRP_SEGS_CAP * sizeof(GrpSeg)));
      if (tab.size() != h_grp_segs.size() || std::memcmp(tab.data(), h_grp_segs.data(), tab.size() * sizeof(GrpSeg)) != 0) {
        h_grp_segs = tab;
        HIP_OK(hipMemcpy(d_grp_segs, h_grp_segs.data(), h_grp_segs.size() * sizeof(GrpSeg), hipMemcpyHostToDevice));
      }
      ka.grp_segs = d_grp_segs; ka.n_grp_segs = static_cast<int32_t>(segs.size());
      last_kernels += std::string(last_kernels.empty() ? "" : ",") + "sampler_grp_all";
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_grp_all), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      hipLaunchKernelGGL(sampler_grp_all, dim3(static_cast<unsigned>(blocks)), dim3(256), lds, st, ka);
      HIP_OK(hipGetLastError());
      all_launched = true;
    }
  }
  // When: the batch's launches outnumber the hardware queues (a whole-gene mix with real read counts: ~20 launches, 243 ->
  // 180 ms per 1500 iterations; a single class's three or four buckets run 3 % faster side by side: 149 vs 155 ms at K = 5;
  // profiles/r03_pe_buckets.txt).  MISO_PE_MULTI=1 forces (tests).
  // ... or some run on 16 / 32 lanes holds several isoform counts: the multi kernel's segments keep them in wavefronts of their own
  bool any_mixed = false;
  for (size_t ri = 0; ri < gen_runs.size(); ri++)
    any_mixed |= p.paired && !gen_runs[ri].wide && (grp_G[ri] == 16 || grp_G[ri] == 32 || (grp_G[ri] == 8 && gen_runs[ri].small)) && fp_rows(gen_runs[ri]) && flat_nc[ri] == 0 && gen_runs[ri].kmin != gen_runs[ri].kmax;
  const bool multi_on = p.paired && !lane_gen && std::getenv("MISO_NO_PE_MULTI") == nullptr && std::getenv("MISO_GENERAL_LANES") == nullptr &&
                        (std::getenv("MISO_PE_MULTI") != nullptr || gen_runs.size() + (n_k2 > 0 ? 1 : 0) > 8 || any_mixed);
  for (size_t r0 = 0; multi_on && !all_launched && r0 < gen_runs.size();) {
    size_t r1 = r0 + 1;
    while (r1 < gen_runs.size() && gen_runs[r1].kc == gen_runs[r0].kc) r1++;
    bool ok = (r1 - r0 >= 2 || (r1 - r0 == 1 && gen_runs[r0].kmin != gen_runs[r0].kmax && (grp_G[r0] == 16 || grp_G[r0] == 32 || (grp_G[r0] == 8 && gen_runs[r0].small)))) &&
              r1 - r0 <= static_cast<size_t>(K2_MAX_SEGS);
    GenRun m;
    m.first = gen_runs[r0].first; m.count = 0; m.kc = gen_runs[r0].kc;
    for (size_t ri = r0; ri < r1 && ok; ri++) {
      const GenRun &run = gen_runs[ri];
      const int G = grp_G[ri];
      ok = flat_nc[ri] == 0 && fp_rows(run) && run.first == m.first + m.count &&
           (run.wide || G == 16 || G == 32 || (G == 64 && run.wave64) || (G == 8 && run.small));
      m.count += run.count; m.kmax = std::max(m.kmax, run.kmax); m.kmin = std::min(m.kmin, run.kmin);
      m.maxq = std::max(m.maxq, run.maxq);
    }
    const GrpShape msh = ok ? grp_shape(m) : GrpShape{0, 0};
    const size_t fp_bytes = fp_bytes_of(m);
    const size_t slice = ok ? grp_slice_bytes(m.kmax, 0, msh.ts) : 0;
    // (the small genes' segment: eight chains per wavefront, their score tables in global memory -- a tstride of its own)
    const size_t slice8 = ok ? grp_slice_bytes(m.kmax, 0, 0) : 0;
    for (size_t ri = r0; ri < r1 && ok; ri++)
      if (!gen_runs[ri].wide) ok = fp_bytes + 4 * static_cast<size_t>(64 / grp_G[ri]) * (grp_G[ri] == 8 ? slice8 : slice) <= LDS_MAX;
    if (ok) {
      KernelArgs ka = a;
      hipStream_t st = stream_for_next();
      ka.slot_event = d_slots + n_k2 + m.first; ka.n_slots = m.count;
      ka.kstride = m.kmax; ka.cstride = 0; ka.tstride = msh.ts;
      ka.pe_dense = 1; ka.pe_force_exact = std::getenv("MISO_PE_FORCE_EXACT") != nullptr;
      size_t lds = 0; int blocks = 0, segs = 0;
      // A run on 16 or 32 lanes per chain is one segment PER ISOFORM COUNT (the run's events are ordered by it): every segment starts
      // a new workgroup, so no wavefront carries chains of two isoform counts -- such a wavefront runs both counts' read loops one after
      // the other (kernels_grp.inl pe_fast), twice a chain's own time, and with the class's longest chains that was the launch's length
      // (round 6, profiles/r06_mix_timeline.txt).  As far as the segments suffice; MISO_PE_NO_KSPLIT=1: a segment per run (A/B).
      auto k_pieces = [&](const GenRun &run) {
        std::vector<int> cuts{0};
        for (int e = 1; e < run.count; e++)
          if (events[h_slots[n_k2 + run.first + e]].K != events[h_slots[n_k2 + run.first + e - 1]].K) cuts.push_back(e);
        cuts.push_back(run.count);
        return cuts;
      };
      size_t want_segs = 0;
      for (size_t ri = r0; ri < r1; ri++)
        want_segs += (gen_runs[ri].wide || grp_G[ri] == 64) ? 1 : k_pieces(gen_runs[ri]).size() - 1;
      const bool ksplit = want_segs <= static_cast<size_t>(K2_MAX_SEGS) && std::getenv("MISO_PE_NO_KSPLIT") == nullptr;
      for (size_t ri = r0; ri < r1; ri++) {
        GenRun &run = gen_runs[ri];
        const long chains = static_cast<long>(run.count) * p.noChains;
        if (run.wide) {
          ka.seg_slot[segs] = run.first - m.first; ka.seg_block[segs] = blocks;
          const size_t lds0 = align_up(fp_bytes + 4 * slice, 16);
          ka.red_off = static_cast<int32_t>(lds0);
          lds = std::max(lds, lds0 + 96);
          blocks += static_cast<int>(wide_setup(run, chains, st));
          ka.coop_tab = run.d_coop_tab; ka.coop_mem = run.d_coop_mem;
          ka.seg_ts[segs] = msh.ts;
          ka.seg_lanes[segs++] = K2_WIDE;
        } else {
          const int G = grp_G[ri], cpw = 64 / G;
          lds = std::max(lds, fp_bytes + 4 * static_cast<size_t>(cpw) * (G == 8 ? slice8 : slice));
          const std::vector<int> cuts = (ksplit && G != 64) ? k_pieces(run) : std::vector<int>{0, run.count};
          for (size_t c = 0; c + 1 < cuts.size(); c++) {
            const long pc = static_cast<long>(cuts[c + 1] - cuts[c]) * p.noChains;
            ka.seg_slot[segs] = run.first - m.first + cuts[c]; ka.seg_block[segs] = blocks;
            blocks += static_cast<int>(((pc + cpw - 1) / cpw + 3) / 4);
            ka.seg_ts[segs] = G == 8 ? 0 : msh.ts;
            ka.seg_lanes[segs++] = G;
          }
        }
        in_multi[ri] = 1;
      }
      ka.seg_slot[segs] = m.count; ka.seg_block[segs] = blocks; ka.n_segs = segs;
      last_kernels += std::string(last_kernels.empty() ? "" : ",") + "sampler_grp_multi<" + std::to_string(m.kc) + ">";
#define MISO_GRP_MULTI(KC)                                                                                   \
  {                                                                                                          \
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sampler_grp_multi<KC>),                       \
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));          \
    hipLaunchKernelGGL((sampler_grp_multi<KC>), dim3(static_cast<unsigned>(blocks)), dim3(256), lds, st, ka); \
  }
      switch (m.kc) {
      case 4: MISO_GRP_MULTI(4) break;
      case 8: MISO_GRP_MULTI(8) break;
      case 12: MISO_GRP_MULTI(12) break;
      case 16: MISO_GRP_MULTI(16) break;
      default: MISO_GRP_MULTI(32) break;
      }
#undef MISO_GRP_MULTI
      HIP_OK(hipGetLastError());
    }
    r0 = r1;
  }
  run_in_multi = in_multi;
  for (size_t ri = 0; ri < gen_runs.size(); ri++) if (gen_runs[ri].wide && !run_lane[ri] && !in_multi[ri]) launch_gen_run(ri);
  if (n_k2w > 0 && k2w_multi) {
    lanes_per_chain = k2w_plan.seg_lanes[k2w_plan.n_segs - 1];
    last_kernels = "sampler_k2_multi<2, " + std::to_string(k2w_plan.wpb) + ">";
    launch_k2_multi(a, stream_for_next(), true);
  } else if (n_k2w > 0) {
    lanes_per_chain = k2w_G;
    last_kernels = k2_name(k2w_G, true);
    launch_k2(a, k2w_G, stream_for_next(), true);
  }
  if (n_k2 - n_k2w > 0) {
    lanes_per_chain = k2_G;
    if (lane_route) {   // collapsed single-end batch: one chain per lane, no read loop (kernels_lane.hip)
      KernelArgs ka = a;
      ka.slot_event = d_slots; ka.n_slots = n_k2; ka.pair_waves = 0;
      ka.logfact = d_logfact; ka.tstride = logfact_n;   // (tstride: the table's entries, for sampler_lane_ilp's LDS copy)
      const long chains = static_cast<long>(n_k2) * p.noChains;
      const int G = lane_G;
      lanes_per_chain = G;
      last_kernels += std::string(last_kernels.empty() ? "" : ",") + (G == 1 ? std::string(lane_ilp ? "sampler_lane_ilp" : "sampler_lane") : "sampler_k2c<" + std::to_string(G) + ">");
      const unsigned grid = static_cast<unsigned>(((chains + 64 / G - 1) / (64 / G) + 3) / 4);
      hipStream_t st = stream_for_next();
      switch (G) {
      case 1: {
        unsigned wgs = static_cast<unsigned>((chains + 255) / 256);
        {   // fewer wavefronts than SIMDs (or than two per SIMD): spread the chains over all of them (kernels_lane.hip lane_body)
          const long simds = wave_slots / 2, waves = (chains + 63) / 64;
          const char *sp = std::getenv("MISO_LANE_SPREAD");   // 0: never; n: n wavefronts per SIMD (experiments)
          const int per_simd = sp ? std::atoi(sp) : 1;
          if (lane_ilp && per_simd > 0 && waves < simds * per_simd && chains >= 64) {
            const long target = simds * per_simd;
            const int cpw = static_cast<int>((chains + target - 1) / target);
            if (cpw >= 8 && cpw < 64) { ka.pair_waves = cpw; wgs = static_cast<unsigned>(((chains + cpw - 1) / cpw + 3) / 4); }
          }
        }
        if (lane_ilp) hipLaunchKernelGGL(sampler_lane_ilp, dim3(wgs), dim3(256), 0, st, ka);
        else hipLaunchKernelGGL(sampler_lane, dim3(wgs), dim3(256), 0, st, ka);
        break;
      }
      case 2: hipLaunchKernelGGL(sampler_k2c<2>, dim3(grid), dim3(256), 0, st, ka); break;
      case 4: hipLaunchKernelGGL(sampler_k2c<4>, dim3(grid), dim3(256), 0, st, ka); break;
      default: hipLaunchKernelGGL(sampler_k2c<8>, dim3(grid), dim3(256), 0, st, ka); break;
      }
      HIP_OK(hipGetLastError());
    } else if (k2_multi) {
      lanes_per_chain = k2_plan.seg_lanes[k2_plan.n_segs - 1];
      last_kernels += std::string(last_kernels.empty() ? "" : ",") + k2m_name;
      launch_k2_multi(a, stream_for_next());
    } else if (k2_mix > 0) {
      last_kernels += std::string(last_kernels.empty() ? "" : ",") + k2_mix_name(k2_G);
      launch_k2_mix(a, k2_G, stream_for_next());
    } else {
      last_kernels += std::string(last_kernels.empty() ? "" : ",") + k2_name(k2_G, false);
      launch_k2(a, k2_G, stream_for_next());
    }
  }
  for (size_t ri = 0; ri < gen_runs.size(); ri++) if (!gen_runs[ri].wide && !run_lane[ri] && !in_multi[ri]) launch_gen_run(ri);
  for (size_t i = 1; i < kernel_no && i <= aux_streams.size() && std::getenv("MISO_SERIAL_KERNELS") == nullptr; i++) {
    HIP_OK(hipEventRecord(aux_done[i - 1], aux_streams[i - 1]));
    HIP_OK(hipStreamWaitEvent(stream, aux_done[i - 1], 0));
  }
  HIP_OK(hipEventRecord(ev1, stream));
  start_clock_probe();
  launched = true; launched_once = true; downloaded = false; summarized = false; compared = false;
}

// The clock probe of this launch (clock_probe_kernel above).  First the flag's store, behind the launch's last kernel on the
// batch's stream; then the probe on a stream of its own: on a hardware queue of its own it starts at once, beside the
// sampler kernels.  Should its stream share the batch's queue (or a profiler serialise the dispatches) it starts when
// everything before it is done, finds the flag set and reports a window of no length, which sync() discards -- in no
// order of execution does it wait for something queued behind it.  Give-up time: 1.5 x the last launch's kernel time + 20 ms.
void miso_batch::start_clock_probe() {
  probe_armed = false;
  if (!clock_probe || probe_failed) return;
  if (!probe_stream) {
    HIP_OK(hipStreamCreateWithFlags(&probe_stream, hipStreamNonBlocking));
    HIP_OK(hipMalloc(&d_probe, 8 * sizeof(unsigned long long)));
    HIP_OK(hipMemset(d_probe, 0, 8 * sizeof(unsigned long long)));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) wall_khz = khz;
  }
  probe_gen++;
  uint32_t *flag = reinterpret_cast<uint32_t *>(d_probe + 7);
  hipLaunchKernelGGL(clock_probe_stop, dim3(1), dim3(64), 0, stream, flag, probe_gen);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamWaitEvent(probe_stream, ev0, 0));
  const double cap_ms = 1.5 * (last_ms > 0.f ? last_ms : 2000.0) + 20.0;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, probe_stream, flag, probe_gen, d_probe,
                     static_cast<unsigned long long>(cap_ms * wall_khz));
  HIP_OK(hipGetLastError());
  probe_armed = true;
}

void miso_batch::read_clock_probe() {
  last_clock_ghz = 0.0; last_probe_ms = 0.0;
  if (!probe_armed) return;
  probe_armed = false;
  HIP_OK(hipStreamSynchronize(probe_stream));
  unsigned long long h[5];
  HIP_OK(hipMemcpy(h, d_probe, sizeof(h), hipMemcpyDeviceToHost));
  if (h[4] != 0) { probe_failed = true; return; }   // it gave up: never again for this batch (each time would cost its give-up time)
  const double ms = static_cast<double>(h[2] - h[0]) / wall_khz;
  // a window much shorter than the launch: the probe ran behind the kernels, not beside them
  if (ms < 0.5 * last_ms || ms <= 0.0) return;
  last_probe_ms = ms;
  last_clock_ghz = static_cast<double>(h[3] - h[1]) / (ms * 1e6);
}

void miso_batch::sync(float *ms) {
  if (!launched) MISO_FAIL(MISO_EINVAL, "batch not launched");
  HIP_OK(hipSetDevice(device));
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipEventElapsedTime(&last_ms, ev0, ev1));
  if (ms) *ms = last_ms;
  read_clock_probe();
  // chains on several workgroups: did any group give up waiting for its members (coop.hpp)?
  bool gave_up = false;
  for (const K2Coop *cc : {&k2_coop_se, &k2w_coop}) {
    if (!cc->d_mem || cc->chains == 0) continue;
    std::vector<uint32_t> w(static_cast<size_t>(cc->chains) * COOP_WORDS);
    HIP_OK(hipMemcpy(w.data(), cc->d_mem, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int c = 0; c < cc->chains; c++) gave_up |= w[static_cast<size_t>(c) * COOP_WORDS + 1] != 0;
  }
  for (const GenRun &run : gen_runs) {
    if (!run.d_coop_mem || run.coop_chains == 0) continue;
    std::vector<uint32_t> w(static_cast<size_t>(run.coop_chains) * COOP_WORDS);
    HIP_OK(hipMemcpy(w.data(), run.d_coop_mem, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int c = 0; c < run.coop_chains; c++) gave_up |= w[static_cast<size_t>(c) * COOP_WORDS + 1] != 0;
  }
  if (!gave_up) { converge_rounds(ms); return; }
  // A chain's workgroups did not all become resident in time (a busy device: other batches, other processes).  The
  // batch must not fail for it -- the reference's workers share nothing (misopy/miso.py:165-187) --, so the launch is
  // repeated here, in this process, with every chain on ONE workgroup: same results bit for bit (the contract's random
  // numbers do not depend on the layout), no exchange between workgroups left that could wait.
  if (no_coop) MISO_FAIL(MISO_EINTERNAL, "a chain timed out waiting for its workgroups although every chain has one workgroup");
  const float first_ms = last_ms;
  no_coop = true; coop_retries++;
  std::fill(coop_n.begin(), coop_n.end(), 1);
  for (GenRun &run : gen_runs) {
    if (run.d_coop_tab) (void) hipFree(run.d_coop_tab);
    if (run.d_coop_mem) (void) hipFree(run.d_coop_mem);
    run.d_coop_tab = nullptr; run.d_coop_mem = nullptr; run.coop_tab.clear(); run.coop_chains = 0;
  }
  for (K2Coop *cc : {&k2_coop_se, &k2w_coop}) {
    if (cc->d_tab) (void) hipFree(cc->d_tab);
    if (cc->d_mem) (void) hipFree(cc->d_mem);
    *cc = K2Coop{};
  }
  k2_plan_key = k2w_plan_key = -1; coop_wgs_used = 0;
  pool_cleared = false;   // (the abandoned chains left their sample columns half written)
  std::fprintf(stderr, "[miso] a chain on several workgroups gave up waiting for its workgroups after %.1f ms (busy device?): "
                       "re-running the launch with one workgroup per chain\n", first_ms);
  launch(last_seed, last_first_event_id);
  sync(ms);
}

// stop = CONVERGENT_MEAN: what the reference does after every round of its iteration loop (miso.c:903-925,
// miso_paired.c:501-523).  Events whose chains have not converged -- and whose schedule is still below maxIterations --
// run another round, noIterations' = 3 noIterations - 2 noBurnIn with noBurnIn' = noIterations, and of that round's samples
// the LAST noSamples replace the first round's (miso.c:976-983), here in the device pool, so that the summaries, the file
// writer and the getters see one finished batch.  The reference CONTINUES its chains: round r keeps the iterations
// [G_r + B_r, G_r + N_r) of one chain, G_r = the iterations of all earlier rounds.  The device keeps no chain state
// between launches, so it runs iterations [0, G_r + N_r) from the start with burn-in G_r + B_r: iteration m of a chain
// draws from the same addresses whatever the schedule, so the first G_r iterations repeat the earlier rounds bit for bit
// and the kept window is the reference's (round 6; rounds 4 - 5 ran N_r from the start and kept [B_r, N_r): the same law
// in the limit, half the burn-in).  The reference's loop counter restarts with every round and so does its rule "no
// proposal terms in a round's first ratio" (miso.c:866): the kernels are told where the later rounds open
// (KernelArgs::round_tab, device.hpp RoundOpen).  The CPU checker's counter mode continues its chains like the
// reference and addresses the draws by the chain's own iteration number: equal bit for bit (tests/test_gpu_convergent.py).
// The next round is a batch of its own (only the unconverged events; its sync() recurses).  A launch reproduces at most
// MISO_MAX_ROUNDS rounds (every round at least doubles the kept window: 2^7 x the first); beyond that the last stands.
void miso_batch::converge_rounds(float *ms) {
  // once per launch: a second sync() must not test the samples the further rounds have already put in place (they might
  // pass now and reset the accept counts' bookkeeping, or fail and run the rounds -- and the paired-end sums -- twice)
  if (converged_done) { if (ms) *ms = last_ms; return; }
  converged_done = true;
  rounds = 1;
  iters_counted.clear(); went_on.clear();
  // this round's own schedule, the reference's (noIterations, noBurnIn): the batch's for the first round
  const int N = round_iters > 0 ? round_iters : p.noIterations, B = round_iters > 0 ? round_burn : p.noBurnIn;
  if (p.stop != MISO_STOP_CONVERGENT_MEAN || p.maxIterations <= N || events.empty()) return;
  if (static_cast<int>(round_starts.size()) >= MISO_MAX_ROUNDS - 1) return;
  const int S0 = S(), C = p.noChains;
  if (S0 < C) return;                        // fewer kept samples than chains: nothing to assess
  std::vector<unsigned char> out(out_bytes);
  HIP_OK(hipMemcpy(out.data(), d_out, out_bytes, hipMemcpyDeviceToHost));
  std::vector<int> again;
  for (size_t i = 0; i < events.size(); i++)
    if (!convergent_mean(reinterpret_cast<const double *>(out.data() + h_events[i].off_samples), events[i].K, C, S0))
      again.push_back(static_cast<int>(i));
  if (again.empty()) return;
  const int64_t N2 = 3LL * N - 2LL * B, total = static_cast<int64_t>(p.noIterations) + N2;
  if (total > INT32_MAX) return;             // (the kernels count iterations in 32 bits)
  miso_params_t p2 = p;
  p2.noIterations = static_cast<int>(total); p2.noBurnIn = p.noIterations + N;   // miso.c:921-923, behind what has been run
  p2.want_counts_trace = 0; p2.device_match = 0;
  std::unique_ptr<miso_batch> next(batch_new(p2));
  next->round_iters = static_cast<int>(N2); next->round_burn = N;
  next->round_starts = round_starts; next->round_starts.push_back(p.noIterations);
  for (int i : again) {
    next->events.push_back(events[i]);
    const bool pinned = i < static_cast<int>(event_ids.size()) && event_ids[i] >= 0;
    next->event_ids.push_back(pinned ? event_ids[i] : static_cast<int64_t>(last_first_event_id + static_cast<uint32_t>(i)));
  }
  next->collapsed = collapsed; next->collapsed_level = collapsed_level; next->no_coop = no_coop;
  float next_ms = 0.f;
  next->upload(device);
  next->launch(last_seed, 0);
  next->sync(&next_ms);
  const int Sn = next->S();
  std::vector<unsigned char> nout(next->out_bytes);
  HIP_OK(hipMemcpy(nout.data(), next->d_out, next->out_bytes, hipMemcpyDeviceToHost));
  // iterations behind an event's accept count: the single-end loop starts both counters afresh every round, the
  // paired-end one never does (miso.c:847 against miso_paired.c:345, 453)
  iters_counted.assign(events.size(), p.paired ? p.noIterations : N);
  for (size_t j = 0; j < again.size(); j++) {
    const int i = again[j];
    const DevEvent &d = h_events[i], &n = next->h_events[j];
    const PackedEvent &e = events[i];
    const size_t skip = static_cast<size_t>(Sn - S0);
    HIP_OK(hipMemcpyAsync(d_out + d.off_samples, next->d_out + n.off_samples + skip * e.K * 8, static_cast<size_t>(S0) * e.K * 8,
                          hipMemcpyDeviceToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_out + d.off_loglik, next->d_out + n.off_loglik + skip * 8, static_cast<size_t>(S0) * 8,
                          hipMemcpyDeviceToDevice, stream));
    if (e.n_draw > 0)
      HIP_OK(hipMemcpyAsync(d_out + d.off_drawass, next->d_out + n.off_drawass, e.n_draw, hipMemcpyDeviceToDevice, stream));
    // accept counts.  The next launch counted from the chain's start: that IS the paired-end number; the single-end one
    // is what the last round added, i.e. minus this launch's own count (the same chain up to here).  An event that went
    // through further rounds in `next` comes back already settled.
    const bool settled = !next->iters_counted.empty() && next->event_went_on(static_cast<int>(j));
    ChainStats *mine = reinterpret_cast<ChainStats *>(out.data() + d.off_stats);
    const ChainStats *theirs = reinterpret_cast<const ChainStats *>(nout.data() + n.off_stats);
    for (int c = 0; c < C; c++) {
      const int32_t before = mine[c].accepted;
      mine[c] = theirs[c];
      if (!p.paired && !settled) mine[c].accepted -= before;
    }
    iters_counted[i] = !next->iters_counted.empty() ? next->iters_counted[j] : (p.paired ? total : N2);
    HIP_OK(hipMemcpyAsync(d_out + d.off_stats, mine, sizeof(ChainStats) * C, hipMemcpyHostToDevice, stream));
  }
  went_on.assign(events.size(), 0);
  for (int i : again) went_on[i] = 1;
  HIP_OK(hipStreamSynchronize(stream));
  rounds = 1 + next->rounds;
  last_ms += next_ms;
  if (ms) *ms = last_ms;
  coop_retries += next->coop_retries;
  last_kernels += "," + next->last_kernels;
}

// Posterior mean and Chen-Shao credible interval of every isoform, computed where the samples are
// (credible_intervals.py:31-55: order statistics int(round(alpha/2 n)) - 1 and
// int(round((1 - alpha/2) n)) - 1, Python-2 rounding = half away from zero).
void miso_batch::summarize(double confidence_level, bool as_text) {
  if (!launched) MISO_FAIL(MISO_EINVAL, "batch not launched");
  // stop = CONVERGENT_MEAN: the launch is only finished once sync() has run its further rounds; without them the pool holds
  // the unconverged first round
  if (p.stop == MISO_STOP_CONVERGENT_MEAN && !converged_done) sync(nullptr);
  HIP_OK(hipSetDevice(device));
  const int n = static_cast<int>(events.size()), Sn = S();
  const double alpha = 1 - confidence_level;
  const int lo = static_cast<int>(std::floor((alpha / 2) * Sn + 0.5)) - 1;
  const int hi = static_cast<int>(std::floor((1 - alpha / 2) * Sn + 0.5)) - 1;
  if (!(lo > 0 && hi > 0 && hi < Sn))   // the reference asserts both indices > 0
    MISO_FAIL(MISO_EINVAL, "Too few samples for a credible interval");
  h_sum_off.assign(n, 0);
  uint64_t off = 0; int kmax = 1;
  for (int i = 0; i < n; i++) { h_sum_off[i] = off; off += 3 * events[i].K; kmax = std::max(kmax, events[i].K); }
  h_summary.assign(off, 0.0);
  if (n == 0) { summarized = true; return; }
  uint64_t *d_off = nullptr; double *d_sum = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_off), n * sizeof(uint64_t)));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_sum), off * sizeof(double)));
  HIP_OK(hipMemcpyAsync(d_off, h_sum_off.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(summarize_kernel, dim3(n, kmax), dim3(256), 0, stream, d_events, d_out, n, Sn, lo, hi,
                     d_off, d_sum, as_text ? 1 : 0);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(h_summary.data(), d_sum, off * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  (void) hipFree(d_off); (void) hipFree(d_sum);
  summarized = true;
}

// Two-sample comparison of this batch (sample 1) with `other` (sample 2), event by event and
// index-paired (hypothesis_test.py:89-179): both must hold the same events in the same order.
void miso_batch::compare(miso_batch &other, double smoothing) {
  if (!launched || !other.launched) MISO_FAIL(MISO_EINVAL, "batch not launched");
  if (device != other.device) MISO_FAIL(MISO_EINVAL, "Batches to compare live on different devices");
  if (events.size() != other.events.size() || S() != other.S())
    MISO_FAIL(MISO_EINVAL, "Batches to compare differ in events or samples per event");
  if (!(smoothing > 0)) MISO_FAIL(MISO_EINVAL, "Invalid smoothing parameter");
  const int n = static_cast<int>(events.size()), Sn = S();
  if (Sn < 2) MISO_FAIL(MISO_EINVAL, "Too few samples to compare");
  std::vector<uint64_t> off(n);
  uint64_t tot = 0; int kmax = 1;
  for (int i = 0; i < n; i++) {
    if (events[i].K != other.events[i].K) MISO_FAIL(MISO_EINVAL, "Events to compare differ in isoforms");
    off[i] = tot; tot += 4 * events[i].K; kmax = std::max(kmax, events[i].K);
  }
  h_compare.assign(tot, 0.0);
  if (n == 0) { compared = true; return; }
  HIP_OK(hipSetDevice(device));
  HIP_OK(hipStreamSynchronize(other.stream));
  uint64_t *d_off = nullptr; double *d_cmp = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_off), n * sizeof(uint64_t)));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_cmp), tot * sizeof(double)));
  HIP_OK(hipMemcpyAsync(d_off, off.data(), n * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(compare_kernel, dim3(n, kmax), dim3(256), 0, stream, d_events, d_out, other.d_events,
                     other.d_out, n, Sn, smoothing, d_off, d_cmp);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(h_compare.data(), d_cmp, tot * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  (void) hipFree(d_off); (void) hipFree(d_cmp);
  compared = true;
}

// A batch that holds posterior samples produced elsewhere -- parsed `.miso` files (summarize_miso / compare_miso
// work on directories of them, misopy/samples_utils.py:263-329, hypothesis_test.py:186-345): event i has K[i] isoforms
// and S samples in the file's layout (rows of K values).  Only the output pool exists; summarize / compare and their
// getters work on it as on a sampled batch.
void miso_batch::adopt_samples(int n, const int *K, int Sn, const double *const *samples, int dev) {
  if (uploaded) release();
  if (device_count() <= 0) MISO_FAIL(MISO_ENODEVICE, "no HIP device: the summaries have no CPU path");
  if (n < 0 || Sn < 1) MISO_FAIL(MISO_EINVAL, "Invalid number of events or samples");
  device = dev;
  HIP_OK(hipSetDevice(dev));
  events.assign(n, PackedEvent{});
  h_events.assign(n, DevEvent{});
  uint64_t off = 0;
  for (int i = 0; i < n; i++) {
    if (K[i] < 1 || K[i] > MISO_MAX_ISOFORMS) MISO_FAIL(MISO_EINVAL, "Invalid number of isoforms");
    events[i].K = K[i];
    DevEvent &d = h_events[i];
    d.K = K[i];
    d.off_samples = off; off = align_up(off + static_cast<uint64_t>(Sn) * K[i] * 8, 16);
    d.off_trace = NO_TRACE; d.off_dense = d.off_sfixd = NO_DENSE;
  }
  out_bytes = std::max<uint64_t>(off, 16);
  HIP_OK(hipStreamCreate(&stream));
  HIP_OK(hipEventCreate(&ev0));
  HIP_OK(hipEventCreate(&ev1));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_events), std::max<size_t>(n, 1) * sizeof(DevEvent)));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d_out), out_bytes));
  if (n) HIP_OK(hipMemcpy(d_events, h_events.data(), n * sizeof(DevEvent), hipMemcpyHostToDevice));
  std::vector<unsigned char> h(out_bytes, 0);
  for (int i = 0; i < n; i++) std::memcpy(h.data() + h_events[i].off_samples, samples[i], static_cast<size_t>(Sn) * K[i] * 8);
  HIP_OK(hipMemcpy(d_out, h.data(), out_bytes, hipMemcpyHostToDevice));
  n_k2 = n_k2w = n_gen = 0; gen_runs.clear();
  uploaded = launched = true; downloaded = false; summarized = compared = false;
}

void miso_batch::download() {
  if (!launched) MISO_FAIL(MISO_EINVAL, "batch not launched");
  // stop = CONVERGENT_MEAN: the launch is only finished once sync() has run its further rounds; without them the pool holds
  // the unconverged first round
  if (p.stop == MISO_STOP_CONVERGENT_MEAN && !converged_done) sync(nullptr);
  HIP_OK(hipSetDevice(device));
  h_out.resize(out_bytes);
  HIP_OK(hipMemcpy(h_out.data(), d_out, out_bytes, hipMemcpyDeviceToHost));
  downloaded = true;
}

namespace miso {

miso_batch *batch_new(const miso_params_t &p) {
  validate_params(p);
  auto b = std::make_unique<miso_batch>();
  b->p = p;
  if (p.paired) b->fd = normal_fragment(p.normalMean, p.normalVar, p.numDevs, p.readLength);
  return b.release();
}

void selftest_detmath(const double *x, int n, double *e, double *l, double *s, double *q) {
  if (device_count() <= 0) MISO_FAIL(MISO_ENODEVICE, "no HIP device");
  double *d[5];
  for (auto &ptr : d) HIP_OK(hipMalloc(reinterpret_cast<void **>(&ptr), std::max(n, 1) * 8));
  HIP_OK(hipMemcpy(d[0], x, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(selftest_detmath_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], n, d[1],
                     d[2], d[3], d[4]);
  HIP_OK(hipGetLastError());
  HIP_OK(hipDeviceSynchronize());
  double *outs[4] = {e, l, s, q};
  for (int i = 0; i < 4; i++) HIP_OK(hipMemcpy(outs[i], d[i + 1], n * 8, hipMemcpyDeviceToHost));
  for (auto ptr : d) (void) hipFree(ptr);
}

void selftest_philox(const uint32_t *in6, int n, uint32_t *out4) {
  if (device_count() <= 0) MISO_FAIL(MISO_ENODEVICE, "no HIP device");
  uint32_t *din, *dout;
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&din), std::max(n, 1) * 24));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&dout), std::max(n, 1) * 16));
  HIP_OK(hipMemcpy(din, in6, n * 24, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(selftest_philox_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, din, n, dout);
  HIP_OK(hipGetLastError());
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy(out4, dout, n * 16, hipMemcpyDeviceToHost));
  (void) hipFree(din); (void) hipFree(dout);
}

}  // namespace miso

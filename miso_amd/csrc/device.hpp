// device.hpp -- HBM layout shared by the host runtime and the HIP kernels.
#pragma once

#include <cstddef>
#include <cstdint>

namespace miso {

// Per-event constants, doubles, at DevEvent::off_consts:
//   [0, K)      cst_k    single-end: log(effective length) (miso.c:136-138)
//                        paired-end: assscores_k           (miso_paired.c:412-419)
//   [K, 2K)     iscore_k single-end: -log(l_k)             (miso.c:781-783); paired: unused
//   [2K, 3K)    hyper_k - 1                                  (miso.c:174)
//   [3K + 0]    lgamma(sum hyper)   [3K + 1] sum lgamma(hyper_k)   (miso.c:172-178)
//   [3K + 2]    sigma = 0.2/K^2     [3K + 3] proposal sd           (miso.c:328, 188)
//   [3K + 4]    (2 pi sigma)^(-(K-1)/2)                            (miso.c:101)
constexpr int CONST_EXTRA = 5;

struct DevEvent {
  int32_t K;
  int32_t n_draw;      // reads with >= 2 compatible isoforms
  int32_t n_reads;
  int32_t base_bad;    // paired: a fixed read carries a non-finite score
  int64_t base_sfix;   // paired: fixed reads' score sum, 2^-26 fixed point (miso_philox.h)
  // byte offsets into the input pool
  uint64_t off_consts; // double[3K + CONST_EXTRA]
  uint64_t off_base;   // int32[K]: reads with exactly one compatible isoform, per isoform
  uint64_t off_draw;   // SE: uint32 mask[n_draw (padded to 4)]; PE: uint16 frag[n_draw x K]
  uint64_t off_cls;    // SE: class table, CLS_WORDS uint32 per drawing-read class + a sentinel row
  uint64_t off_clsmask;// SE: uint16 (class << 8 | member) per threshold of the class path
  int32_t n_dcls;      // SE: number of drawing-read classes (0 = use the masks)
  int32_t n_units;     // SE: number of work units (Philox blocks x classes touching them)
  int32_t max_cls;     // SE: most isoforms any drawing class is compatible with
  int32_t n_pairs;     // SE: sum over classes of (isoforms - 1)
  uint32_t explicit_id;// the event's id in the Philox counter when has_id (else first_event_id + index)
  int32_t has_id;
  uint64_t off_sfix;   // PE: int32[K x il] fixed-point scores, MISO_SFIX_BAD = non-finite
  int32_t pe_delta;    // PE, K = 2: no drawing read touches a non-finite score (sampler_k2 MODE 2)
  int32_t dense_nobad; // PE dense records: no drawing read touches a non-finite score
  uint64_t off_dense;  // PE, 3 <= K <= PE_DENSE_KMAX: quad records of pe_dense (kernels_grp.inl), NO_DENSE = none
  uint64_t off_units;  // SE with a class table: u32 per work unit (Philox block x class), in unit order:
                       // bits 0-3 the block's words that belong to the class, 4-11 the class, 12-31 the block
  uint64_t off_sfixd;  // PE dense: int32[K x (il + 2)] scores in the records' index space
  // byte offsets into the output pool
  uint64_t off_samples; // double[S x K]  (reference layout: K x S column-major)
  uint64_t off_loglik;  // double[S]
  uint64_t off_drawass; // uint8[n_draw]: chain 0's final pick for every drawing read
  uint64_t off_stats;   // per chain: {uint64 counts_hash; int32 accepted; int32 pad}
  uint64_t off_trace;   // int32[(M+1) x C x K] or ~0 when not requested
  // algorithm = MARGINAL / CLASSES (single-end; sampler_marginal, kernels_marginal.hip): the classes the marginal
  // likelihood sums over, K + 1 doubles each: a weight per isoform and the number of reads.  MARGINAL: every read
  // class with a compatible isoform, in the column order of the header's classes, weight 1 / effective length for its
  // isoforms (miso.c:800-808), 0 for the others; CLASSES: the gene's possible read classes that have reads, weight =
  // the class's share of the isoform's read start positions (miso.c:788-803)
  uint64_t off_mcls;
  int32_t n_mcls;
  int32_t pad_mcls;
};
constexpr int MARGINAL_VECTORS = 9;   // sampler_marginal: alpha, alpha', psi, psi', log psi (2), log(psi / last) (2), scratch
inline std::size_t marginal_lds_bytes(int ks, int lanes) { return static_cast<std::size_t>(MARGINAL_VECTORS) * ks * lanes * 8; }

// one event of match_kernel (kernels_match.hip)
struct MatchEvent {
  int32_t K;
  int32_t n_reads;     // reads (single-end) or pairs (paired-end)
  int32_t exidx_off;   // K + 1 exon offsets of the event, relative to ex_off
  int32_t ex_off;      // the event's exon coordinates
  int32_t read_off;    // first read (mate) of the event in the batch-wide read arrays
  int32_t out_off;     // first output slot: masks[out_off + r] / frags[(out_off + r) * K + k]
};

#ifdef __HIPCC__
#define MISO_DEVHOST_EARLY __host__ __device__
#else
#define MISO_DEVHOST_EARLY
#endif

struct ChainStats {
  uint64_t counts_hash;
  int32_t accepted;
  uint32_t hw_id;      // HW_REG_HW_ID of the wavefront that ran the chain (placement diagnostics)
};

constexpr int FLAT_WIDE = 0x100;  // sampler_flat wave_tab: the workgroup's four wavefronts share ONE chain
constexpr int K2_MAX_SEGS = 16;
constexpr int MISO_MAX_ROUNDS = 8;   // stop = CONVERGENT_MEAN: rounds a device launch can reproduce (every round at least doubles the kept window)
constexpr int K2_WIDE = 512;   // seg_lanes value: one chain per workgroup
constexpr int K2_RED_BYTES = 2 * 8 * 16 + 16 + 16;   // two buffers x (up to) 8 wavefronts x {int64 score sum, int count, int bad} + the barrier's flag + the psi a workgroup-wide chain's first four wavefronts publish (kernels_k2.inl)

// sampler_grp_all (kernels_grp_all.hip): one segment of the launch -- the run of one isoform-count class, sixteen lanes per chain
struct GrpSeg {
  int32_t block0;           // the segment's first workgroup (the table ends with a sentinel entry: block0 = the grid)
  int32_t slot0, n_slots;   // its events in the launch's list
  int32_t kc;               // the class: 4, 8, 12, 16, 32
  int32_t kstride, tstride; // the run's slice layout (KernelArgs::kstride, tstride)
};

struct KernelArgs {
  const DevEvent *events;
  const unsigned char *in_pool;
  unsigned char *out_pool;
  const double *frag_prob;  // paired: normalised fragment-length probabilities [il]
  int32_t il;
  const int32_t *slot_event;  // this launch's events (indices into `events`), n_slots of them
  int32_t n_slots;
  int32_t kstride;          // sampler_grp: isoform stride of the per-chain LDS slices
  int32_t cstride;          // sampler_grp: SE drawing-read classes per chain in the LDS slice (0 = no class path)
  int32_t tstride;          // sampler_grp PE: score-table entries per chain staged in LDS (0 = none)
  int32_t nc;               // sampler_flat: chains per wavefront
  int32_t n_events;
  int32_t C, M, B, lag;     // chains, iterations (incl. burn-in), burn-in, lag
  int32_t start;            // MISO_START_AUTO / MISO_START_UNIFORM
  uint32_t first_event_id;
  int32_t pair_waves;       // sampler_k2 with 8 wavefronts per workgroup: wavefronts w and w + 4 (one SIMD)
                            // take the heaviest and the lightest remaining chain group (runtime.hip)
  int32_t pe_dense;         // sampler_grp PE: every event of the launch has dense records (pe_dense) and the LDS
                            // probability table carries the two extra entries PE_ZERO, PE_ONE (0 = the quad loops of pe_quads)
  int32_t pe_force_exact;   // tests: every read through pe_dense's exact (cold) scan; the single-end two-isoform loop: every lane rescans its blocks for high halves on the threshold
  int32_t flat_desc;        // sampler_flat: read loop over the unit descriptors (flat_units_desc), 0 = the walking loop
  int32_t flat_thr_skip;    // sampler_flat: thresholds only for the chains whose psi changed in the last Metropolis-Hastings step
  int32_t mix_blocks;       // sampler_k2_mix<GA, GB>: the first mix_blocks workgroups run the first mix_slots events
  int32_t mix_slots;        // with GA lanes per chain, the rest the remaining events with GB (runtime.hip)
  const int32_t *wave_tab;  // sampler_flat: two words per wavefront: first chain of the launch's list, chains | FLAT_WIDE (runtime.hip)
  const int32_t *coop_tab;  // workgroup-wide chains on SEVERAL workgroups (coop.hpp): per workgroup {chain (slot of the launch's
                            // list), rank among the chain's workgroups, their number, the chain's scratch index}; null = one each
  uint32_t *coop_mem;       // COOP_WORDS dwords of exchange scratch per cooperative chain, zeroed before the launch
  uint32_t coop_max_polls;  // polls after which a cooperative workgroup gives up on its chain's other workgroups (0 = COOP_MAX_POLLS)
  const double *logfact;    // sampler_lane: log(k!) for k = 0 .. the batch's largest number of drawing reads (miso_binomial.h)
  int32_t red_off;          // sampler_k2_multi: byte offset of the workgroup-wide chains' reduction scratch in the dynamic LDS
  uint64_t seed;
  // sampler_k2_multi (kernels_k2m.hip): the launch's events (ordered by drawing reads, most first) cut into runs
  // of equal lanes per chain.  Run s: workgroups [seg_block[s], seg_block[s + 1]), events (slots)
  // [seg_slot[s], seg_slot[s + 1]) of the launch's list, seg_lanes[s] lanes per chain (K2_WIDE = the whole workgroup).
  // stop = CONVERGENT_MEAN (runtime.hip converge_rounds): the iterations that open a later round of the reference's loop,
  // ascending, MISO_MAX_ROUNDS - 1 entries in global memory, unused ones -1; null for a launch without later rounds
  // (every launch of stop = FIXEDNO)
  const int32_t *round_tab;
  // sampler_k2_multi<0, 8>, one round: the two wavefronts of a SIMD keep step by priority (kernels_k2.inl k2_balance)
  int32_t balance;
  int32_t wide_dedup;       // sampler_k2 WIDE with eight wavefronts: the Metropolis-Hastings step on the first four only (kernels_k2.inl)
  const GrpSeg *grp_segs;   // sampler_grp_all: n_grp_segs segments + the sentinel, in global memory
  int32_t n_grp_segs;
  int32_t n_segs;
  int32_t seg_block[K2_MAX_SEGS + 1];
  int32_t seg_slot[K2_MAX_SEGS + 1];
  int32_t seg_lanes[K2_MAX_SEGS];
  int32_t seg_ts[K2_MAX_SEGS];   // sampler_grp_multi: the segment's tstride (eight chains of a wavefront keep their score tables in global memory)
};

#ifdef __HIPCC__
// The reference's loop counter starts at 0 in every round of stop = CONVERGENT_MEAN (miso.c:845 `for (m=0, ...` inside
// `while (1)`), and with it the rule that a round's first Metropolis-Hastings ratio leaves the proposal terms out
// (miso.c:866 `m > 0 ? 1 : 0`).  A launch that re-runs a chain from its start through round r (runtime.hip
// converge_rounds) is told where the later rounds open; at() is called once per iteration, in order.
struct RoundOpen {
  int next, i;
  __device__ explicit RoundOpen(const KernelArgs &a) : next(a.round_tab ? a.round_tab[0] : -1), i(0) {}
  __device__ __attribute__((always_inline)) bool at(const KernelArgs &a, int m) {
    if (m == 0) return true;
    if (__builtin_expect(m != next, 1)) return false;
    // (a table in global memory behind a pointer that may be null: nothing the compiler can load ahead of the loop and keep
    // in scalar registers -- seven entries among the kernel's by-value arguments were, and came back through v_readlane in
    // every iteration of the kernels that are short of scalar registers; a run-time index into the by-value arguments makes
    // the compiler keep a copy of them in scratch)
    i++;
    next = i < MISO_MAX_ROUNDS - 1 ? a.round_tab[i] : -1;
    return true;
  }
};
#endif

#ifdef __HIPCC__
// Launches of several rounds of workgroups (KernelArgs::balance == 2; experiment, round 6): a SIMD's arbiter serves its
// OLDEST wavefront whenever it can issue, so wavefronts run nearly one after the other and the launch ends with every
// SIMD's youngest running alone.  A wavefront's priority falls with its progress instead -- by quarters of its iterations:
// a newcomer catches up with its SIMD's residents, the residents stay within a quarter of each other and finish together.
__device__ __attribute__((always_inline)) inline void prio_by_progress(const KernelArgs &a, int m) {
  if (a.balance != 2) return;
  const int q = a.M >> 2;
  if (m == 0) __builtin_amdgcn_s_setprio(3);
  else if (m == q) __builtin_amdgcn_s_setprio(2);
  else if (m == 2 * q) __builtin_amdgcn_s_setprio(1);
  else if (m == 3 * q) __builtin_amdgcn_s_setprio(0);
}
#endif

constexpr uint64_t NO_TRACE = ~0ull;
constexpr uint64_t NO_DENSE = ~0ull;
// Paired-end dense records (host.cpp pack_event_masks -> kernels_grp.inl pe_dense), available when the
// fragment-length range has at most 254 values (il + 2 <= 256: mean +- 4 sd up to sd = 31; wider ranges take
// the plain records' loops).  One BYTE per (read, isoform): the fragment-length index f, or PE_ZERO = il for an
// incompatible isoform (probability -0.0: never picked), or PE_ONE = il + 1 for isoform 0 of a padding read
// (probability 1: always picked, score 0) -- the LDS probability table has il2 = il + 2 entries, the event's
// score table (sfix_dense) K rows of il2.  One quad = 4 reads x K bytes (byte h = j K + k), then one dword of
// flags (bit j: read j has MORE than two compatible isoforms): K + 1 dwords.  What a chain streams per
// iteration is half of the u16 records' bytes -- the loop is bound by what crosses from the infinity cache into
// the L2s (~5 TB/s, profiles/r02_pe_k5_summary.txt).  After the event's last quad comes one more made of
// padding reads only (the lanes beyond the end work on it).
constexpr int PE_DENSE_KMAX = 20;
// Two isoforms (sampler_k2 MODE 2), same idea: one u32 per read, f0 | (il + f1) << 16, into tables of
// 2 il + 2 entries -- probabilities [fp, fp, 0.0, 1.0], scores [isoform 0, isoform 1, 0, 0] -- so that one index
// serves the probability and the score; a padding read is (2 il) | (2 il + 1) << 16: weight 0 against a
// positive one, never isoform 0, score 0.  Quads of four reads, one quad of padding reads after the last.
MISO_DEVHOST_EARLY inline int pe_k2_entries(int il) { return 2 * il + 2; }
MISO_DEVHOST_EARLY inline int pe_dense_il2(int il) { return il + 2; }
MISO_DEVHOST_EARLY inline int pe_dense_quad_dwords(int K) { return K + 1; }
constexpr int MAX_DRAW_CLASSES = 64;  // single-end: per-class integer thresholds up to this many classes (if the LDS slice fits)

// LDS bytes of one chain's slice in sampler_grp (layout: kernels_grp.hip `carve`): isoform stride ks
// (even), cs single-end drawing-read classes (0 = no class path), ts paired-end score entries.
#ifdef __HIPCC__
#define MISO_DEVHOST __host__ __device__
#else
#define MISO_DEVHOST
#endif
// class table row: {mask, first unit, first unit - first block, head | tail << 4 word masks};
// after the sentinel row: A_k = reads of the classes whose last isoform is <= k, k < K
constexpr int CLS_WORDS = 4;
MISO_DEVHOST inline int grp_cls_bytes(int ks, int cs) {
  return cs > 0 ? cs * (ks - 1) * 4 + ((cs * (ks - 1)) & 1) * 4 + (CLS_WORDS * (cs + 1) + ks + (ks & 1)) * 4 +
                      ((cs * (ks - 1) * 2 + 7) & ~7)
                : 0;
}
MISO_DEVHOST inline int grp_slice_bytes(int ks, int cs, int ts) {
  return 15 * ks * 8 + 32 + (3 * ks + (ks & 1)) * 4 + grp_cls_bytes(ks, cs) + ((ts + 1) & ~1) * 4;
}
// ---- sampler_flat (kernels_flat.inl): one chain's LDS slice, byte offsets.  ks = isoform stride
// (the launch's largest K), cs = most drawing-read classes of any event of the launch. ----
constexpr int FLAT_SX = 16;     // per-chain double scalars
constexpr int FLAT_MISC = 24;   // per-chain int scalars
struct FlatLayout {
  int psi, alpha, lp, tb, lr;   // double[2][ks]: buffer `parity` = current state and its cached logs, the other = proposal
  int tc, u2;                   // double[ks] scratch
  int cst, isc, hm1;            // double[ks] per-isoform constants (device.hpp, top)
  int sx;                       // double[FLAT_SX]
  int cnt, bas, dl;             // int[ks]: picks of the drawing reads, fixed reads, D_k
  int misc;                     // int[FLAT_MISC]
  int thr;                      // u32[cs x flat_trow(ks)]: thr[c][k] = words below it pick an isoform <= k (0 beyond K - 1)
  int ctab;                     // u32[CLS_WORDS x (cs + 1) + ks]: class rows, sentinel row, A_k
  int bytes;
};
// width of the read loop's register rows for a largest isoform count ks: the instantiated widths of
// kernels_flat.inl's flat_units (2, 3 | 4..7 | 9, 11 | 15 | 19, 23, 31); thr rows have this stride
MISO_DEVHOST inline int flat_trow(int ks) {
  const int tw = ks - 1;
  return tw <= 3 ? (tw <= 2 ? 2 : 3) : (tw <= 7 ? (tw <= 4 ? 4 : tw) : (tw <= 9 ? 9 : (tw <= 11 ? 11 : (tw <= 15 ? 15 : (tw <= 19 ? 19 : (tw <= 23 ? 23 : 31))))));
}
MISO_DEVHOST inline FlatLayout flat_layout(int ks, int cs) {
  FlatLayout L{};
  const int tr = flat_trow(ks), kd = ks > tr + 1 ? ks : tr + 1;
  int o = 0;
  L.psi = o; o += 16 * ks; L.alpha = o; o += 16 * ks; L.lp = o; o += 16 * ks; L.tb = o; o += 16 * ks;
  L.lr = o; o += 16 * ks;
  L.tc = o; o += 8 * ks; L.u2 = o; o += 8 * ks;
  L.cst = o; o += 8 * ks; L.isc = o; o += 8 * ks; L.hm1 = o; o += 8 * ks;
  L.sx = o; o += 8 * FLAT_SX;
  L.cnt = o; o += 4 * ks; L.bas = o; o += 4 * ks; L.dl = o; o += 4 * kd;
  L.misc = o; o += 4 * FLAT_MISC;
  L.thr = o; o += 4 * cs * tr;
  L.ctab = o; o += 4 * (CLS_WORDS * (cs + 1) + ks);
  L.bytes = (o + 15) & ~15;
  return L;
}
constexpr uint16_t FRAG_NONE = 0xFFFF;
constexpr int32_t SFIX_BAD = INT32_MIN;  // == MISO_SFIX_BAD

}  // namespace miso

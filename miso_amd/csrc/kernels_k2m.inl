// kernels_k2m.inl -- two-isoform events, one launch, a lane width PER EVENT (sampler_k2_multi).
//
// The reference's cost per event is O(reads) with nothing shared between events (miso.c:845-900, one process per
// chunk of events, misopy/miso.py:165-187); real read counts span 20 ... 10^5 per event.  A launch with one lane
// width for every chain (sampler_k2) puts a 50 000-draw chain on the same three or four lanes as a 200-draw one and
// the whole launch waits for it.  Here the host cuts the launch's event list (ordered by drawing reads, most first)
// into runs of equal lanes per chain -- 1 ... 64, or the whole workgroup for the largest events (k2_body's WIDE) --
// so that every wavefront's Gibbs step takes about the same time (runtime.hip: plan_k2), and this kernel runs the
// body of sampler_k2 that belongs to its workgroup's run.  One launch because concurrent launches are placed at the
// dispatcher's whim (see sampler_k2_mix); workgroups are numbered heaviest run first, so the hardware's in-order
// dispatch is longest-processing-time-first when the launch needs more than one round of resident wavefronts.
// Results do not depend on the lanes per chain (the arithmetic contract): same tests.
// WPB = wavefronts per workgroup: 8 when the whole launch is resident at once (one workgroup per CU, the two
// wavefronts of a SIMD = a heavy and a light one, a.pair_waves); fewer when the launch needs several rounds -- a
// workgroup can only start when ALL its wavefronts' slots are free, so with 8 the slots of early finishers idle.
// Instantiated one per translation unit (kernels_k2m_*.hip): a dozen bodies each.
#include "kernels_k2.inl"

namespace miso {

// NARROW: the plan's runs all have one or two lanes per chain (MISO's default settings on like-sized events: 240 000 chains on
// one lane each).  Only those two bodies, so that the kernel fits the 168 registers of THREE wavefronts per SIMD -- which the
// several-rounds layout is worth more than registers (199.0 ms against 206.9 ms at two, same bodies) -- without asking the
// wider bodies to spill for it: their Metropolis-Hastings step keeps both outcomes' proposals in flight since round 6 and needs
// 177 (hg19-like read counts at the default settings: 131.3 ms at two wavefronts per SIMD, 137.5 ms at three with 36 bytes of
// scratch, 137.6 ms in round 5).  __launch_bounds__'s second argument is the minimum wavefronts per SIMD.
template <int MODE, int WPB, bool NARROW = false>
__global__ __launch_bounds__(64 * WPB, NARROW ? 3 : 2) void sampler_k2_multi(const KernelArgs a) {
  // One-round single-end launches (a.wave_tab): behind the workgroup-wide chains' a.mix_blocks workgroups every WAVEFRONT
  // looks up which run's wavefront it is -- the host pairs the launch's wavefronts by estimated duration ACROSS the runs,
  // heaviest with lightest on one SIMD (wavefronts w and w + 4), so that every SIMD carries the same work (runtime.hip
  // k2_pair_table).  Single-end bodies have no workgroup-level step: the eight wavefronts of a workgroup may belong to
  // different runs.
  if constexpr (MODE == 0 && WPB == 8) {
    if (a.wave_tab != nullptr && static_cast<int>(blockIdx.x) >= a.mix_blocks) {
      const int e = __builtin_amdgcn_readfirstlane(a.wave_tab[(static_cast<size_t>(blockIdx.x) - a.mix_blocks) * 8 + (threadIdx.x >> 6)]);
      if (e < 0) return;
      const int sg = e >> 20;
      const long wid = e & 0xFFFFF;
      KernelArgs part = a;
      part.slot_event = a.slot_event + a.seg_slot[sg];
      part.n_slots = a.seg_slot[sg + 1] - a.seg_slot[sg];
      part.pair_waves = 0;
      switch (a.seg_lanes[sg]) {
#define MISO_K2M_WAVE(GG) case GG: k2_body<GG, 0, 8>(part, 0u, 0u, wid); break;
      MISO_K2M_WAVE(1) MISO_K2M_WAVE(2) MISO_K2M_WAVE(3) MISO_K2M_WAVE(4) MISO_K2M_WAVE(5) MISO_K2M_WAVE(6)
      MISO_K2M_WAVE(8) MISO_K2M_WAVE(10) MISO_K2M_WAVE(12) MISO_K2M_WAVE(16) MISO_K2M_WAVE(32) MISO_K2M_WAVE(64)
#undef MISO_K2M_WAVE
      default: break;
      }
      return;
    }
  }
  int s = 0;
  while (s + 1 < a.n_segs && static_cast<int>(blockIdx.x) >= a.seg_block[s + 1]) s++;
  KernelArgs part = a;
  part.slot_event = a.slot_event + a.seg_slot[s];
  part.n_slots = a.seg_slot[s + 1] - a.seg_slot[s];
  const unsigned bx = blockIdx.x - static_cast<unsigned>(a.seg_block[s]);
  const unsigned gx = static_cast<unsigned>(a.seg_block[s + 1] - a.seg_block[s]);
  if constexpr (NARROW) {
    if (a.seg_lanes[s] == 1) k2_body<1, MODE, WPB>(part, bx, gx);
    else if (a.seg_lanes[s] == 2) k2_body<2, MODE, WPB>(part, bx, gx);
    return;
  }
  switch (a.seg_lanes[s]) {
#define MISO_K2M_CASE(GG) case GG: k2_body<GG, MODE, WPB>(part, bx, gx); break;
#define MISO_K2M_CASE_SE(GG) case GG: if constexpr (MODE == 0) k2_body<GG, MODE, WPB>(part, bx, gx); break;
  // paired-end: powers of two from 4 (the chains' score tables bound the chains per wavefront, runtime.hip)
  MISO_K2M_CASE_SE(1) MISO_K2M_CASE_SE(2) MISO_K2M_CASE_SE(3) MISO_K2M_CASE(4) MISO_K2M_CASE_SE(5) MISO_K2M_CASE_SE(6)
  MISO_K2M_CASE(8) MISO_K2M_CASE_SE(10) MISO_K2M_CASE_SE(12) MISO_K2M_CASE(16) MISO_K2M_CASE(32) MISO_K2M_CASE(64)
  // (no 21: that body alone needs 177 vector registers, and the kernel's allocation -- the largest of its bodies --
  // decides whether a third wavefront fits a SIMD when the launch runs in several rounds: <= 168)
#undef MISO_K2M_CASE
#undef MISO_K2M_CASE_SE
  default:   // K2_WIDE
    if constexpr (WPB >= 2) k2_body<64, MODE, WPB, true>(part, bx, gx);
    break;
  }
}

}  // namespace miso

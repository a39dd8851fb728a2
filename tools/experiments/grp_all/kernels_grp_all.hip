// sampler_grp_all: a whole-gene paired-end batch (3 - 20 isoforms per gene, BASELINE configs[3]) in ONE launch.
//
// The reference runs every gene by itself, one after the other in a worker's list (misopy/run_miso.py:205-256,
// miso_paired.c:451-498): genes share nothing, so the order is free.  Round 3 gave every isoform-count class its own
// kernel (sampler_grp<G, true, KC>, register budgets by class) and round 4 measured what that costs a mix: five kernels
// side by side in the hardware queues, placed at the dispatcher's whim (15.7 - 17.6 k genes/s from box to box), the
// longest class -- 17 to 20 isoforms -- still running when the others' workgroups are gone.  All paired-end classes
// have the same register budget (two workgroups per CU), so their bodies fit ONE kernel: the batch's runs (class x
// size bucket) are segments of one grid, numbered longest chains first ACROSS the classes, and the hardware's
// in-order dispatch is the schedule.  Same bodies, same bits (tests/test_gpu_heavy_tail.py, MISO_PE_ALL=1).
#include "kernels_grp.inl"

namespace miso {

template <int KC> __device__ __forceinline__ void grp_all_class(const KernelArgs &b, unsigned blk, int lanes) {
  switch (lanes) {
  case K2_WIDE: grp_body<64, true, KC, true>(b, blk); break;
  case 64: grp_body<64, true, KC, false>(b, blk); break;
  case 32: grp_body<32, true, KC, false>(b, blk); break;
  default: grp_body<16, true, KC, false>(b, blk); break;
  }
}

__global__ __launch_bounds__(256, MISO_GRP_PE_BLOCKS) void sampler_grp_all(const KernelArgs a) {
  int s = 0;
  while (s + 1 < a.n_segs && static_cast<int>(blockIdx.x) >= a.seg_block[s + 1]) s++;
  s = __builtin_amdgcn_readfirstlane(s);
  KernelArgs b = a;
  b.slot_event = a.slot_event + a.seg_slot[s];
  b.n_slots = a.seg_slot[s + 1] - a.seg_slot[s];
  b.kstride = a.seg_ks[s]; b.tstride = a.seg_ts[s]; b.red_off = a.seg_red[s];
  b.coop_tab = a.seg_coop_tab[s]; b.coop_mem = a.seg_coop_mem[s];
  const unsigned blk = blockIdx.x - static_cast<unsigned>(a.seg_block[s]);
  const int lanes = a.seg_lanes[s];
  switch (a.seg_kc[s]) {
  case 4: grp_all_class<4>(b, blk, lanes); break;
  case 8: grp_all_class<8>(b, blk, lanes); break;
  case 12: grp_all_class<12>(b, blk, lanes); break;
  case 16: grp_all_class<16>(b, blk, lanes); break;
  default: grp_all_class<32>(b, blk, lanes); break;
  }
}

}  // namespace miso

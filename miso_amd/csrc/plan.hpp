// plan.hpp -- lanes per chain, event by event, for the two-isoform sampler's one-launch-many-widths kernels
// (kernels_k2m.hip).  Pure host arithmetic (no HIP): tests/test_plan.py drives it through miso_plan_lanes.
//
// The reference spends O(reads) per event and events share nothing (miso.c:845-900); its dispatcher hands every
// worker process a chunk of events (misopy/miso.py:165-187).  Here the unit that must not straggle is the
// wavefront: a wavefront holds 64 / G chains and its Gibbs step lasts as long as its longest chain's read loop.
#pragma once

#include <cstdint>
#include <vector>

#include "device.hpp"

namespace miso {

// What one wavefront pays per Gibbs step, in VALU issue slots (time x issue rate at two wavefronts per SIMD, measured
// with forced single-width launches of 300- and 1500-read events, tools/sweep_multi.py, profiles/r03_k2_cost_model.txt):
// step[min(G, 4)] for everything that does not depend on the reads (Metropolis-Hastings, threshold, reduction,
// recording) + `block` per Philox block (four draws) a lane works through.
struct LaneCost {
  double step[5] = {0, 0, 0, 0, 0};   // index 1..4 = lanes cooperating on the scalar step
  double block = 48;
  int uq = 2;              // blocks per lane and trip of the read loop (trips are whole)
  bool paired = false;     // paired-end loop: 2 x trips + 1 blocks; single-end: uq x trips (+ 1 for a partial quad)
  int coop_min_quads = 8;  // a chain on several workgroups keeps at least this many blocks per lane (tests lower it)
  int draws = 4;           // draws per Philox block: four words; single-end EIGHT half-words (miso_philox.h, lazy low bits)
  // exact: what the kernels' loops do (trips are whole, a partial block costs a block)
  double blocks_per_lane(int lanes, int n_draw) const {
    const int nfq = n_draw / draws;
    const int trips = (nfq + uq * lanes - 1) / (uq * lanes);
    return paired ? 2.0 * trips + 1.0 : static_cast<double>(uq) * trips + ((n_draw % draws) ? 1.0 : 0.0);
  }
  double wave_step(int lanes, int n_draw) const {   // lanes = lanes striding over the chain's draws (K2_WIDE: 64 x wpb)
    return step[lanes >= 4 ? 4 : lanes] + block * blocks_per_lane(lanes, n_draw);
  }
  // smooth in n_draw (no rounding to whole trips): what the widths are chosen by, so that the choice never flips
  // back and forth between neighbouring events of the ordered list
  double smooth_step(int lanes, int n_draw) const {
    return step[lanes >= 4 ? 4 : lanes] + block * (n_draw / (static_cast<double>(draws) * lanes) + 0.5 * uq + (paired ? 1.0 : 0.5));
  }
};

// The measured models (profiles/r03_k2_cost_model.txt).  Single-end: step = MH + threshold + reduction + recording
// with 1, 2, 3, >= 4 lanes sharing the transcendentals; block = one Philox4x32 block + four compares.  Paired-end
// (MODE 2, dense records): block = generator + four reads' weights, compares and score gathers.
// (round 4: the generator has 7 rounds instead of 10, include/miso_philox.h: three rounds x four instructions fewer per block
// than the 52 / 115 measured in round 3)
// (single-end since the lazy low bits: a block is EIGHT reads -- 24 generator instructions + 5 per word for the packed
// below / equal arithmetic + the loop's share: 108 per two blocks; the launch time hardly moves between 44 and 96,
// profiles/r04_lazy_low_bits.txt)
inline LaneCost k2_cost_single() {
  LaneCost c; c.step[1] = 1750; c.step[2] = 1170; c.step[3] = 840; c.step[4] = 720; c.block = 54; c.uq = 2; c.paired = false;
  c.draws = 8;
  return c;
}
inline LaneCost k2_cost_paired() {
  LaneCost c; c.step[1] = c.step[2] = c.step[3] = c.step[4] = 650; c.block = 103; c.uq = 2; c.paired = true;
  return c;
}

struct LanePlan {
  int n_segs = 0;
  int32_t seg_block[K2_MAX_SEGS + 1] = {0};   // first workgroup of run s (and the total at n_segs)
  int32_t seg_slot[K2_MAX_SEGS + 1] = {0};    // first event (slot of the launch's list) of run s
  int32_t seg_lanes[K2_MAX_SEGS] = {0};       // lanes per chain of run s; K2_WIDE = one chain per workgroup
  long waves = 0;            // wavefronts launched
  int rounds = 1;            // 1: every workgroup is resident at once; 2: more workgroups than the device holds
  double target = 0;         // the bound on a wavefront's Gibbs step the widths were chosen for
  double est_total = 0;      // sum over wavefronts of wave_step
  double est_max = 0;        // largest wave_step
  double est_last = 0;       // wave_step of the last (lightest) wavefront
  double est_pair = 0;       // one round: the busiest SIMD = heaviest + lightest wavefront of a run (a.pair_waves)
  int wpb = 8;               // wavefronts per workgroup the plan was made for
  double est = 0;            // the launch's estimated duration in VALU issue slots of one SIMD (what plan_lanes minimises)
  std::vector<int> wide_wgs; // per event of the K2_WIDE run (the first, if any): workgroups per chain (coop.hpp), 1 = its own only
};

// n_draw: the launch's events' drawing reads, most first; `chains` chains per event; widths: the instantiated lanes
// per chain, ascending; wide_wpb: wavefronts of a workgroup-wide chain (0 = not available); wpb: wavefronts per
// workgroup; resident_wgs: workgroups the device (or this kernel's share of it) holds at once; max_cpw: most
// chains per wavefront the kernel's LDS allows (64 = no limit); coop_max: most workgroups of one workgroup-wide chain;
// coop_budget: most workgroups all chains on several workgroups may take together.
LanePlan plan_lanes(const int *n_draw, int n_events, int chains, const int *widths, int n_widths, int wide_wpb,
                    int wpb, int resident_wgs, int max_cpw, const LaneCost &cost, double forced_target = 0.0,
                    int coop_max = 1, int coop_budget = 1 << 30);

}  // namespace miso

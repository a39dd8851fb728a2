// batch.hpp -- the batch object behind miso_batch_t (definition shared by runtime.hip / capi.hip)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include <functional>

#include "device.hpp"
#include "host.hpp"
#include "plan.hpp"

namespace miso {
int device_count();
void set_device(int d);
miso_batch *batch_new(const miso_params_t &p);
int choose_lanes_per_chain(long chains, int max_quads, int wave_slots, int max_cpw);
void selftest_detmath(const double *x, int n, double *e, double *l, double *s, double *q);
void selftest_philox(const uint32_t *in6, int n, uint32_t *out4);
}  // namespace miso

struct miso_batch {
  miso_params_t p{};
  miso::FragmentDist fd;                 // paired only
  std::vector<miso::PackedEvent> events;
  std::vector<int64_t> event_ids;        // per event: explicit Philox event id, -1 = first_event_id + index
  // device
  int device = -1;
  bool uploaded = false, launched = false, downloaded = false;
  bool pool_cleared = false;      // the output pool has been zeroed since the upload
  bool launched_once = false;     // since the upload
  std::vector<int> coop_n;        // per event: workgroups of a workgroup-wide paired-end chain (upload: by its share of the batch's work)
  int coop_wgs_used = 0;          // cooperative workgroups handed out to this batch's wide runs (<= coop_gen_budget)
  int coop_gen_budget = 0;        // ... of the COOP_MAX_WGS a launch may hold in all; the two-isoform plans share the rest (launch())
  bool no_coop = false;           // set by sync() after a chain on several workgroups gave up: every chain on one workgroup from now on
  int coop_retries = 0;           // launches sync() had to repeat that way (miso_batch_coop_retries)
  uint64_t last_seed = 0;         // what the last launch() was called with (sync()'s re-run)
  uint32_t last_first_event_id = 0;
  // stop = CONVERGENT_MEAN (miso.c:903-925): sync() runs the events that have not converged again on the longer
  // schedule (runtime.hip converge_rounds) and puts the tail of their samples where the first round's were
  std::vector<int64_t> iters_counted;   // per event: iterations behind its accept count (empty: noIterations each)
  // a batch that runs a LATER round (made by converge_rounds): that round's own schedule, the reference's noIterations /
  // noBurnIn at that point (p holds the device's: everything from the chain's start), and the iterations at which the
  // rounds after the first open (KernelArgs::round_tab)
  int round_iters = 0, round_burn = 0;
  std::vector<int> round_starts;
  miso::GrpSeg *d_grp_segs = nullptr;     // sampler_grp_all's segment table on the device (KernelArgs::grp_segs), h_grp_segs its host copy
  std::vector<miso::GrpSeg> h_grp_segs;
  int32_t *d_round_tab = nullptr;   // ... on the device (KernelArgs::round_tab)
  std::vector<char> went_on;      // per event: it ran a further round in the last launch's converge_rounds
  bool event_went_on(int i) const { return i < static_cast<int>(went_on.size()) && went_on[i] != 0; }
  int rounds = 1;                 // rounds the last launch took (1 = the events' own schedule sufficed)
  int prio_lo = 0, prio_hi = 0;   // the device's stream priority range as this batch uses it (equal: priorities off)
  bool converged_done = false;    // stop = CONVERGENT_MEAN: this launch's further rounds have run (sync() is idempotent; launch() clears it)
  void converge_rounds(float *ms);
  bool coop_enabled() const;      // chains may use several workgroups (coop.hpp): not after a time-out, not with MISO_NO_COOP=1
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // the shader clock of a launch (miso_batch_set_clock_probe; runtime.hip clock_probe_kernel)
  bool clock_probe = false, probe_armed = false, probe_failed = false;
  hipStream_t probe_stream = nullptr;
  unsigned long long *d_probe = nullptr;   // {t0 real, t0 cycles, t1 real, t1 cycles, gave up, -, -, flag}
  uint32_t probe_gen = 0;                  // the flag's value that ends the current launch's probe
  double wall_khz = 100000.0;              // hipDeviceAttributeWallClockRate
  double last_clock_ghz = 0.0, last_probe_ms = 0.0;   // 0: no probe, or its window did not cover the launch
  void start_clock_probe();
  void read_clock_probe();
  std::vector<hipStream_t> aux_streams;   // kernels 2.. of a mixed batch run beside the first
  std::vector<hipEvent_t> aux_done;
  miso::DevEvent *d_events = nullptr;
  unsigned char *d_in = nullptr, *d_out = nullptr;
  double *d_fp = nullptr;
  int32_t *d_slots = nullptr;     // [k2 events sorted by n_draw desc | all other events]
  std::vector<int32_t> h_slots;   // the same list on the host
  bool k2_general = false;        // paired-end, tables too wide for the two-isoform kernel's LDS: K = 2 events take sampler_grp
  bool use_delta = true;          // paired-end: MODE 2 events first in the list (fixed at upload)
  double *d_logfact = nullptr;    // collapsed: log factorials up to the largest two-isoform event's drawing reads
  int logfact_n = 0;
  int collapsed_level = 0;        // 1: two-isoform events; 2: also the events with more isoforms (sampler_lane_k)
  bool collapsed = false;         // single-end two-isoform events: the collapsed Gibbs step (kernels_lane.hip); miso_batch_set_collapsed
  miso::LanePlan k2_plan;         // sampler_k2_multi: the runs of equal lanes per chain (runtime.hip), valid for k2_plan_key
  long k2_plan_key = -1;
  struct K2Coop {                 // a plan's chains on several workgroups (coop.hpp): table, scratch
    int32_t *d_tab = nullptr; uint32_t *d_mem = nullptr; int chains = 0; long key = -1;
  };
  K2Coop k2_coop_se, k2w_coop;
  miso::LanePlan k2w_plan;        // the same for the paired-end MODE 2 events (sampler_k2_multi<2, 4>)
  long k2w_plan_key = -1;
  int n_k2 = 0, n_gen = 0;
  int n_k2w = 0;                  // paired-end: the first n_k2w two-isoform slots take sampler_k2's MODE 2
  // the other events, grouped by isoform-count class (sampler_grp<G, PE, KC> holds K in (KC_prev, KC])
  struct GenRun {
    int first = 0, count = 0;     // slice of d_slots (after the n_k2 two-isoform events)
    int kc = 4;                   // 4, 8, 12, 16 or 32
    int kmax = 2, kmin = 64, maxq = 1;   // most / fewest isoforms, most draw quads
    double sum_q = 0;             // draw quads of all its events (their mean decides the lanes per chain in a batch of several classes)
    int maxcls = 0;               // most drawing-read classes (single-end)
    bool nocls = false;           // some single-end event has no class table (> MAX_DRAW_CLASSES classes)
    bool dense = true;            // paired-end: every event has dense quad records (pe_dense)
    bool small = false;           // paired-end size bucket: genes of few pairs in a batch of several classes -- eight lanes per chain (runtime.hip upload)
    int force_G = 0;              // paired-end size bucket: at least this many lanes per chain (events several times the class's mean size)
    bool wide = false;            // paired-end size bucket: one chain per workgroup (sampler_grp<64, true, KC, true>)
    bool wave64 = false;          // paired-end size bucket: one chain per wavefront (sampler_grp<64, true, KC>)
    // wide runs: which chain every workgroup works on, alone or as one of several (coop.hpp; runtime.hip launch_grp)
    std::vector<int32_t> coop_tab;
    int32_t *d_coop_tab = nullptr;
    uint32_t *d_coop_mem = nullptr;
    int coop_chains = 0;          // chains on more than one workgroup
    int tuned_G = 0;              // lanes per chain picked by the first launch's trial runs
    int tuned_flat = -1;          // sampler_flat (1) or sampler_grp (0) by the first launch's trial runs, -1 = not tried
    // sampler_flat: which chains every wavefront owns (runtime.hip flat_waves; two words per wavefront)
    std::vector<int32_t> wave_tab;
    int32_t *d_wave_tab = nullptr;
    long wave_key = -1;
    int wave_nc = 0;              // most chains of any wavefront = slices per wavefront in LDS
    int wave_wide = 0;            // chains that own a whole workgroup
    bool wave_packed = false;     // wavefronts packed by work units (events of very different sizes)
  };
  std::vector<GenRun> gen_runs;
  int tuned_k2_G = 0;             // ditto for the two-isoform kernel
  // device_match: events whose compatibility is still to be computed (kernels_match.hip)
  struct Pending {
    int event = 0;               // index into `events` (a placeholder until resolved)
    miso::Gene gene;
    std::vector<int> pos;
    miso::CigarTable ct;
    std::vector<double> hyper;   // empty = all ones
  };
  std::vector<Pending> pending;
  float match_ms = 0.f;          // kernel time of the last resolve
  // tests (want_counts_trace): the kernel's raw outputs, per resolved event
  std::vector<std::vector<uint64_t>> kept_masks;   // single-end: N masks
  std::vector<std::vector<uint16_t>> kept_frags;   // paired-end: N x K fragment indices
  void resolve_pending();        // runs match_kernel for all pending events, packs them
  int lanes_per_chain = 0;        // G of the last sampler_k2 launch (0 = none)
  std::string last_kernels;       // names of the kernels of the last launch, comma separated
  std::vector<miso_kernel_stat_t> kernel_stats;   // miso_batch_launch_stats, filled on demand by stats_builder
  // sampler_k2_multi<0, 8>, one round: the launch's wavefronts paired by estimated duration across the runs (runtime.hip)
  std::vector<int32_t> k2_pair_tab;
  int32_t *d_k2_pair_tab = nullptr;
  int k2_pair_wide_blocks = 0, k2_pair_grid = 0;
  std::vector<char> run_in_multi;                 // gen_runs launched as a segment of sampler_grp_multi (1) or sampler_grp_all (2) (last launch)
  std::function<void()> stats_builder;            // set by launch(): the walk over events and wavefronts is not part of a launch
  int wave_slots = 2048;          // resident sampler_k2 wavefronts on the device
  std::vector<miso::DevEvent> h_events;
  std::vector<unsigned char> h_out;
  std::vector<double> h_summary;        // per event: K x {mean, ci_low, ci_high}
  std::vector<uint64_t> h_sum_off;      // offsets (in doubles) into h_summary
  bool summarized = false;
  std::vector<double> h_compare;        // per event: K x {mean1, mean2, bayes factor, posterior density at 0}
  bool compared = false;
  uint64_t in_bytes = 0, out_bytes = 0;
  float last_ms = 0.f;

  int k2_first_event() const {  // the k2 event with the most drawing reads (list is sorted)
    int best = -1;
    for (size_t i = 0; i < events.size(); i++)
      if (events[i].K == 2 && !k2_general && (best < 0 || events[i].n_draw > events[best].n_draw))
        best = static_cast<int>(i);
    return best;
  }
  int S() const { return p.noChains * (p.noIterations - p.noBurnIn) / p.noLag; }
  ~miso_batch() { release(); }
  void release();
  void upload(int dev);
  void launch(uint64_t seed, uint32_t first_event_id);
  void sync(float *ms);
  void download();
  void summarize(double confidence_level, bool as_text = false);
  void adopt_samples(int n, const int *K, int S, const double *const *samples, int dev);
  void compare(miso_batch &other, double smoothing);
};

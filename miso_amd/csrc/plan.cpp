// plan.cpp -- see plan.hpp.
#include "plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace miso {

namespace {

struct Planner {
  const int *nd; int n, C; const int *widths; int nw; int wide_wpb, wpb, resident, max_cpw; const LaneCost &cost;
  int coop_max;   // most workgroups one chain may use (1 = a chain ends at its workgroup)
  int coop_budget;   // most workgroups ALL chains on several workgroups may use together (coop.hpp: resident at once)
  static constexpr double wide_extra = 150.0;   // barrier + LDS round trip of a workgroup-wide chain's reduction
  static constexpr double coop_extra = 1200.0;  // + atomics and the barrier between workgroups (~ 2 us)

  // workgroups of a workgroup-wide chain: the fewest that keep its step within D (coop.hpp), each with at least
  // eight quads per lane
  int wide_wgs(double D, int ndraw) const {
    const int lanes = 64 * wide_wpb;
    if (coop_max <= 1 || cost.smooth_step(lanes, ndraw) + wide_extra <= D) return 1;
    const int cap = std::max(1, std::min(coop_max, (ndraw / cost.draws) / (std::max(1, cost.coop_min_quads) * lanes)));
    for (int m = 2; m <= cap; m++) if (cost.smooth_step(lanes * m, ndraw) + wide_extra + coop_extra <= D) return m;
    return cap;
  }

  // the narrowest width whose wavefront step stays within D; never more lanes than the chain has pairs of draw
  // quads to stride over; the whole workgroup when even 64 lanes overshoot and the chain has work for all of them
  int pick(double D, int ndraw) const {
    const int cap = std::max(1, (ndraw / cost.draws) / 2);
    int last = 0;
    for (int i = 0; i < nw; i++) {
      const int G = widths[i];
      if (64 / G > max_cpw) continue;              // LDS: too many chains per wavefront
      if (last && G > cap) break;
      last = G;
      if (cost.smooth_step(G, ndraw) <= D) return G;
    }
    if (wide_wpb > 0 && last == 64 && ndraw >= 64 * wide_wpb * 16 && cost.smooth_step(64, ndraw) > D) return K2_WIDE;
    return last;
  }
  // the workgroups per chain of a workgroup-wide run's events, in list order (largest first), within the budget
  long wide_list(double D, int first, long events, std::vector<int> *out) const {
    long w = 0, used = 0;
    for (long e = 0; e < events; e++) {
      int m = wide_wgs(D, nd[first + e]);
      if (m > 1 && used + static_cast<long>(m) * C > coop_budget) m = static_cast<int>(std::max<long>(1, (coop_budget - used) / C));
      if (m > 1) used += static_cast<long>(m) * C;
      if (out) out->push_back(m);
      w += static_cast<long>(C) * m;
    }
    return w;
  }
  long wgs_of(int lanes, long events, double D = 0, int first = 0) const {
    const long chains = events * C;
    if (lanes == K2_WIDE) return wide_list(D, first, events, nullptr);
    const int cpw = 64 / lanes;
    return ((chains + cpw - 1) / cpw + wpb - 1) / wpb;
  }
  // runs of equal width along the list for bound D; returns workgroups.  pick() never grows along the list (the
  // drawing reads shrink), so there are at most n_widths + 1 runs; a last run takes whatever is left if the table
  // is full (wider than needed, never narrower)
  long cut(double D, LanePlan *out) const {
    long wgs = 0; int segs = 0;
    int i = 0;
    while (i < n) {
      const int G = pick(D, nd[i]);
      int j = i + 1;
      if (segs == K2_MAX_SEGS - 1) j = n;
      else while (j < n && (nd[j] == nd[j - 1] || pick(D, nd[j]) == G)) j++;
      const long w = wgs_of(G, j - i, D, i);
      if (out) {
        out->seg_slot[segs] = i; out->seg_block[segs] = static_cast<int32_t>(wgs); out->seg_lanes[segs] = G;
        out->seg_slot[segs + 1] = j; out->seg_block[segs + 1] = static_cast<int32_t>(wgs + w);
      }
      wgs += w; segs++;
      i = j;
    }
    if (out) out->n_segs = segs;
    return wgs;
  }
};

}  // namespace

LanePlan plan_lanes(const int *n_draw, int n_events, int chains, const int *widths, int n_widths, int wide_wpb,
                    int wpb, int resident_wgs, int max_cpw, const LaneCost &cost, double forced_target, int coop_max,
                    int coop_budget) {
  LanePlan best;
  if (n_events <= 0 || chains <= 0 || n_widths <= 0) return best;
  const Planner P{n_draw, n_events, chains, widths, n_widths, wide_wpb, wpb, std::max(1, resident_wgs), max_cpw, cost,
                  std::max(1, coop_max), std::max(0, coop_budget)};
  const double wide_extra = Planner::wide_extra;
  // what a bound D costs: the runs, then every wavefront's step (the list is ordered: a wavefront's first chain is
  // its longest)
  auto evaluate = [&](double D) {
    LanePlan plan;
    plan.target = D; plan.wpb = wpb;
    plan.rounds = P.cut(D, &plan) > P.resident ? 2 : 1;
    for (int s = 0; s < plan.n_segs; s++) {
      const int G = plan.seg_lanes[s];
      const long c0 = static_cast<long>(plan.seg_slot[s]) * chains, c1 = static_cast<long>(plan.seg_slot[s + 1]) * chains;
      double first = 0;
      if (G == K2_WIDE) {
        plan.wide_wgs.clear();
        P.wide_list(D, plan.seg_slot[s], plan.seg_slot[s + 1] - plan.seg_slot[s], &plan.wide_wgs);
        for (long c = c0; c < c1; c++) {
          const int m = plan.wide_wgs[static_cast<size_t>(c / chains - plan.seg_slot[s])];
          const double w = cost.wave_step(64 * wide_wpb * m, n_draw[c / chains]) + wide_extra + (m > 1 ? Planner::coop_extra : 0.0);
          plan.est_total += w * wide_wpb * m; plan.est_max = std::max(plan.est_max, w); plan.est_last = w;
          if (c == c0) first = w;
          plan.waves += static_cast<long>(wide_wpb) * m;
        }
        plan.est_pair = std::max(plan.est_pair, 2.0 * first);   // a SIMD's two wavefronts belong to the same chain
        continue;
      }
      const int cpw = 64 / G;
      for (long c = c0; c < c1; c += cpw) {
        const double w = cost.wave_step(G, n_draw[c / chains]);
        plan.est_total += w; plan.est_max = std::max(plan.est_max, w); plan.est_last = w;
        if (c == c0) first = w;
        plan.waves++;
      }
      plan.est_pair = std::max(plan.est_pair, first + plan.est_last);   // the run's heaviest wavefront shares its SIMD with the lightest
    }
    // (the last wavefront's first chain depends on how the runs before it happen to divide into wavefronts: take the
    // smallest event itself, so that the estimate does not jump with the alignment)
    if (plan.n_segs > 0 && plan.seg_lanes[plan.n_segs - 1] != K2_WIDE)
      plan.est_last = cost.wave_step(plan.seg_lanes[plan.n_segs - 1], n_draw[n_events - 1]);
    return plan;
  };
  auto simd_share = [&](const LanePlan &plan) {   // total work over the SIMDs that have any
    const double simds = std::min(0.5 * static_cast<double>(P.resident) * wpb, std::ceil(0.5 * static_cast<double>(plan.waves)));
    return plan.est_total / std::max(1.0, simds);
  };
  if (forced_target > 0) { LanePlan f = evaluate(forced_target); f.est = std::max(simd_share(f), 2.0 * f.est_max); return f; }
  int narrow = widths[0];
  for (int i = 0; i < n_widths; i++) if (64 / widths[i] <= max_cpw) { narrow = widths[i]; break; }
  double lo = cost.step[4] + 2.0 * cost.block;
  double hi = std::max(lo, cost.smooth_step(narrow, n_draw[0])) * 1.001;
  if (P.cut(hi, nullptr) <= P.resident) {
    // The narrowest layout fits the device at once.  Then the best plan is the WIDEST that still does (measured,
    // profiles/r03_k2_target_sweep.txt: 40 000 chains, kernel time against the bound -- the minimum sits at the largest
    // number of wavefronts that is still one round; one more workgroup than the device holds and some SIMDs carry
    // three wavefronts while others carry two: 102 ms -> 146 ms): every workgroup resident from the start, one per
    // CU, the two wavefronts of a SIMD = a heavy and a light one of the same run (a.pair_waves).
    for (int it = 0; it < 60 && hi - lo > 1e-4 * hi; it++) {
      const double mid = 0.5 * (lo + hi);
      if (P.cut(mid, nullptr) <= P.resident) hi = mid; else lo = mid;
    }
    best = evaluate(hi);
    best.est = std::max(simd_share(best), best.est_pair);
    return best;
  }
  // Several rounds.  Time of the launch in VALU issue slots of one SIMD: the hardware starts workgroups in index
  // order as slots free up -- heaviest first, lightest last --, so the launch ends within one of the LAST wavefronts
  // of the ideal, unless a single wavefront outlasts everything else; a wavefront shares its SIMD's issue slots with
  // its neighbours and advances at half speed at best.
  auto estimate = [&](const LanePlan &plan) {
    const double per_simd = plan.est_total / (0.5 * static_cast<double>(P.resident) * wpb);
    return std::max(per_simd + plan.est_last, 2.0 * plan.est_max);
  };
  double best_est = 0;
  bool have = false;
  for (double D = hi; ; D /= 1.05) {     // from the narrowest layout towards the widest
    const LanePlan plan = evaluate(D);
    const double e = estimate(plan);
    if (std::getenv("MISO_PLAN_DEBUG")) std::fprintf(stderr, "[plan] D %.0f waves %ld total %.0f max %.0f last %.0f est %.0f\n", D, plan.waves, plan.est_total, plan.est_max, plan.est_last, e);
    if (!have || e < best_est) { best = plan; best_est = e; best.est = e; have = true; }
    if (D <= lo) break;
  }
  return best;
}

}  // namespace miso

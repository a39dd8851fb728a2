// pel_layout.hpp -- sampler_pel's LDS (kernels_pel.inl; runtime.hip sizes the launch with the same arithmetic).
// Per workgroup: the fragment-length probability row [il + 2 doubles: probabilities, -0.0, 1.0], then one slice per chain.
// A chain's slice, byte offsets: the Metropolis-Hastings vectors (buffer 0 = current state and its cached logs, buffer ks =
// proposal), per-isoform constants, the Gibbs step's hand-over (fixed-point score sum, bad flag, counts), then -- as far as
// the workgroup's LDS allows -- the event's fixed-point score table (ts int32) and the first rd dwords of its dense records.
#pragma once
#include "device.hpp"

namespace miso {

constexpr int PEL_MISC = 12;   // per-chain int scalars
constexpr int PEL_ST = 10;     // >= lane_mh.hpp LANE_MH_ST
enum { PM_K = 0, PM_NDRAW, PM_EV, PM_CHAIN, PM_EVID, PM_RBAD, PM_NQL };
struct PelLayout {
  int psi, alpha, lp, tb, lr;   // double[2][ks]
  int tc, cst, hm1;             // double[ks]
  int st;                       // double[PEL_ST]: the Metropolis-Hastings step's state between iterations (lane_mh.hpp LANE_MH_ST)
  int rfix;                     // int64: base_sfix + the picks' scores of the last Gibbs step
  int cnt, bas, dl;             // int[ks]
  int misc;                     // int[PEL_MISC]
  int stab;                     // int32[ts]
  int rec;                      // uint32[rd]
  int bytes;                    // an odd number of 8-byte words (64 lanes reading one entry of 64 slices: 64 banks)
};
MISO_DEVHOST inline PelLayout pel_layout(int ks, int ts, int rd) {
  PelLayout L{};
  int o = 0;
  L.psi = o; o += 16 * ks; L.alpha = o; o += 16 * ks; L.lp = o; o += 16 * ks; L.tb = o; o += 16 * ks; L.lr = o; o += 16 * ks;
  L.tc = o; o += 8 * ks; L.cst = o; o += 8 * ks; L.hm1 = o; o += 8 * ks;
  L.st = o; o += 8 * PEL_ST;
  L.rfix = o; o += 8;
  L.cnt = o; o += 4 * ks; L.bas = o; o += 4 * ks; L.dl = o; o += 4 * ks;
  L.misc = o; o += 4 * PEL_MISC;
  o = (o + 7) & ~7;
  L.stab = o; o += 4 * ts;
  o = (o + 7) & ~7;
  L.rec = o; o += 4 * rd;
  L.bytes = (o + 7) & ~7;
  if (((L.bytes >> 3) & 1) == 0) L.bytes += 8;
  return L;
}

}  // namespace miso

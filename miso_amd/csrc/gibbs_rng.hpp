// gibbs_rng.hpp -- the Gibbs-site Philox4x32 block (MISO_PHILOX_ROUNDS rounds, include/miso_philox.h) with round 0 split into chain and iteration
// constants (the counter is (block, iteration, site | chain << 8, event): only `block` changes inside
// a read loop).  Output identical to miso_draw_block(seed, event, chain, iter, MISO_SITE_GIBBS, q).
#pragma once
#include <hip/hip_runtime.h>

#include "miso_philox.h"

namespace miso {

struct GibbsRng {
  uint32_t p1lo, p1hi;  // M1 * (site | chain << 8)
  uint32_t c3k1;        // event_id ^ k1
  uint32_t k0, k1;
};

__device__ __forceinline__ GibbsRng gibbs_rng_init(uint64_t seed, uint32_t event_id, uint32_t chain) {
  GibbsRng g;
  g.k0 = static_cast<uint32_t>(seed); g.k1 = static_cast<uint32_t>(seed >> 32);
  const uint64_t p1 = static_cast<uint64_t>(MISO_PHILOX_M1) * (MISO_SITE_GIBBS | (chain << 8));
  g.p1lo = static_cast<uint32_t>(p1); g.p1hi = static_cast<uint32_t>(p1 >> 32);
  g.c3k1 = event_id ^ g.k1;
  return g;
}

// per iteration: n0_round0 = g.p1hi ^ iter ^ g.k0
// REKEY: the nine round keys are wave-uniform, so hipcc hoists all 18 into SGPRs; in the register-
// heavy general kernel they are then spilled to VGPR lanes and fetched back with a v_readlane + s_nop
// per round INSIDE the read loop.  An opaque copy of the key makes them loop-variant: they are
// rebuilt per block with scalar adds, which issue beside the vector work.
template <bool REKEY = false>
__device__ __forceinline__ miso_u32x4 philox_gibbs(const GibbsRng &g, uint32_t q, uint32_t n0_round0) {
  const uint64_t p0 = static_cast<uint64_t>(MISO_PHILOX_M0) * q;
  uint32_t c0 = n0_round0, c1 = g.p1lo, c2 = static_cast<uint32_t>(p0 >> 32) ^ g.c3k1,
           c3 = static_cast<uint32_t>(p0);
  uint32_t gk0 = g.k0, gk1 = g.k1;
  if (REKEY) {   // the key is the launch's seed: uniform, but only readfirstlane proves it to the compiler
    gk0 = __builtin_amdgcn_readfirstlane(gk0); gk1 = __builtin_amdgcn_readfirstlane(gk1);
    asm volatile("" : "+s"(gk0), "+s"(gk1));
  }
  uint32_t k0 = gk0 + MISO_PHILOX_W0, k1 = gk1 + MISO_PHILOX_W1;
#pragma unroll
  for (int r = 1; r < MISO_PHILOX_ROUNDS; r++) {
    const uint64_t a = static_cast<uint64_t>(MISO_PHILOX_M0) * c0;
    const uint64_t b = static_cast<uint64_t>(MISO_PHILOX_M1) * c2;
    const uint32_t n0 = MISO_XOR3(static_cast<uint32_t>(b >> 32), c1, k0);
    const uint32_t n2 = MISO_XOR3(static_cast<uint32_t>(a >> 32), c3, k1);
    c1 = static_cast<uint32_t>(b); c3 = static_cast<uint32_t>(a); c0 = n0; c2 = n2;
    k0 += MISO_PHILOX_W0; k1 += MISO_PHILOX_W1;
  }
  miso_u32x4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

}  // namespace miso

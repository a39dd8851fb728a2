/*
 * miso_philox.h -- the counter-based RNG contract shared by the HIP kernels, the host
 * library and the CPU checker (oracle/miso_oracle.c, counter mode).
 *
 * Why it exists: the reference draws from ONE global sequential stream
 * (pysplicing/src/random.c:491 splicing_rng_default; in CPython, Python's `random`,
 * pyrandom.c:163-184) whose consumption is data dependent (miso.c:65-80 draws only for
 * reads with >= 2 compatible isoforms; miso.c:870 short-circuits the accept draw).  Thousands
 * of events sampled concurrently cannot share such a stream, so every draw gets a fixed
 * address instead:
 *
 *     Philox4x32-7( key = (seed_lo, seed_hi),
 *                    ctr = (block, iteration, site | chain << 8, event_id) ) -> 4 x u32
 *
 *   site MISO_SITE_MH  (0): block 0 word 0        = accept uniform        (miso.c:870)
 *                           word 2+2j, 3+2j       = the two uniforms of proposal normal j
 *                                                   (random.c:1543-1551: u=(int)(2^27 u1)+u2)
 *                           (words continue into block 1, 2, ... for K-1 > 1 normals)
 *   site MISO_SITE_GIBBS (2): word r (block r/4, lane r%4) = the uniform of the r-th read that
 *                           has >= 2 compatible isoforms, counted in read order (miso.c:69-80)
 *                           -- except single-end events with TWO isoforms (LAZY LOW BITS, below): there
 *                           half-word r (block r/8, half r%8; half h = bits 16 (h & 1) .. + 15 of word h / 2)
 *                           is the HIGH half of the read's 32-bit uniform and the same half-word of site
 *   site MISO_SITE_GIBBS_LOW (4)  its LOW half
 *   iteration = m for the main loop (miso.c:847), MISO_ITER_INIT for the set-up draws
 *   (initial proposal miso.c:834 and initial assignment miso.c:841).
 *
 * A uniform is u32 * 2^-32, the same 32-bit resolution as the reference's stand-alone
 * generator (random.c:382 splicing_rng_mt19937_get_real).
 *
 * LAZY LOW BITS (round 4).  A two-isoform single-end read picks isoform 0 iff its uniform u (32 bits) is below a
 * threshold t that is the same for all reads of the chain in that step (miso.c:65-72: rand * sumpsi < cumsum[0]).
 * With u = hi * 2^16 + lo:  u < t  <=>  hi < (t >> 16)  or  (hi == (t >> 16) and lo < (t & 0xFFFF)).  The high halves
 * decide all but one read in 65 536, so the device generates ONE block per EIGHT reads and evaluates the second
 * stream only for a block in which some high half equals t >> 16; the checker assembles both halves for every read.
 * Same 32-bit uniforms, same law, the same picks on both sides bit for bit -- and half the generator work in the loop
 * that is nothing but generator work (24 of 32 instructions per four reads).
 *
 * Philox4x32: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11 -- constants and round
 * function as published there; pinned to the Random123 distribution's known-answer vectors at 7 AND at 10 rounds
 * (tests/test_oracle_golden.py::test_philox_known_answers, tests/test_gpu_contract.py).
 *
 * ROUNDS.  MISO_PHILOX_ROUNDS = 7 (rounds 1 - 3 of this build drew with 10).  The paper's Table 2 gives 7 rounds as the
 * fewest at which Philox4x32 is Crush-resistant -- passes every test of TestU01's SmallCrush, Crush and BigCrush -- and
 * recommends 10 as a default safety margin; Random123 ships philox4x32_R(7, ...) for exactly this use.  The generator
 * the reference itself draws from, MT19937 (random.c:301-448), fails BigCrush's two linear-complexity tests, so 7 rounds
 * is still a stronger generator by that standard than the one whose results are being reproduced.  Why it matters: the
 * read loops are VALU-issue bound and the generator is most of their instructions -- 9 rounds x 4 per block after round 0
 * is hoisted, against 2 (K - 1) x 4 for the compares: 36 of 44 per block for two isoforms; 6 x 4 = 24 of 32 with 7 rounds.
 * What it does not change: every draw's ADDRESS, the bit-exactness of GPU against the CPU checker (both include this
 * header), and the agreement with the reference in law, which bench.py and the tests check row by row with two-sample
 * permutation tests (tests/_dpsi.py).
 */
#ifndef MISO_PHILOX_H
#define MISO_PHILOX_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MISO_HD __host__ __device__ __forceinline__
#else
#define MISO_HD static inline
#endif

#define MISO_SITE_MH    0u
#define MISO_SITE_GIBBS 2u
#define MISO_SITE_GIBBS_LOW 4u   /* (3 = MISO_SITE_COUNTS, miso_binomial.h) */
#define MISO_ITER_INIT  0xFFFFFFFFu

#define MISO_PHILOX_M0 0xD2511F53u
#define MISO_PHILOX_M1 0xCD9E8D57u
#define MISO_PHILOX_W0 0x9E3779B9u
#define MISO_PHILOX_W1 0xBB67AE85u

typedef struct { uint32_t v[4]; } miso_u32x4;

/* a ^ b ^ c.  gfx950 has no v_xor3_b32 but a three-input boolean op, v_bitop3_b32 (truth table
 * 0x96 = parity); hipcc does not form it from two xors, and Philox spends 4 xors per round next to
 * 2 multiplies, so the device code asks for it by name.  Same bits everywhere. */
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__)
#define MISO_XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#else
#define MISO_XOR3(a, b, c) ((a) ^ (b) ^ (c))
#endif

#ifndef MISO_PHILOX_ROUNDS
#define MISO_PHILOX_ROUNDS 7
#endif

/* The version of the COUNTER-MODE CONTRACT: which numbers a seeded run draws and in which order.  The device
   (libmiso_amd.so: miso_contract_version()) and the CPU checker (oracle: orc_contract_version()) are compiled from this
   header and the tests assert that both report the same number, so a stale prebuilt checker -- or results saved by
   another version -- are noticed instead of silently disagreeing.  Bumped whenever a seeded counter-mode
   output can change:
     1  rounds 1 - 3: Philox4x32-10, draws in input order
     4  round 4: Philox4x32-7; lazy low half-words for single-end two-isoform events (MISO_SITE_GIBBS_LOW)
     5  round 5: paired-end draw order by fragment-length rows; division-free binomial inversion (miso_binomial.h)
     6  round 6: stop = CONVERGENT_MEAN continues its chains -- round r keeps the iterations [G_r + B_r, G_r + N_r) of one
        chain addressed by its own iteration number, a round's first ratio without the proposal terms (miso.c:866);
        everything under stop = FIXEDNO as in 5 */
#define MISO_CONTRACT_VERSION 6

/* `rounds` is a compile-time constant at every call site of the product (the loop unrolls); the checker also calls it
   with 10 to pin the implementation to the published vectors of both round counts */
MISO_HD miso_u32x4 miso_philox4x32_r(int rounds, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                     uint32_t k0, uint32_t k1) {
  miso_u32x4 out;
  int r;
#if defined(__clang__)
#pragma unroll
#endif
  for (r = 0; r < rounds; r++) {
    uint64_t p0 = (uint64_t) MISO_PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t) MISO_PHILOX_M1 * c2;
    uint32_t n0 = MISO_XOR3((uint32_t) (p1 >> 32), c1, k0);
    uint32_t n2 = MISO_XOR3((uint32_t) (p0 >> 32), c3, k1);
    c1 = (uint32_t) p1;
    c3 = (uint32_t) p0;
    c0 = n0;
    c2 = n2;
    k0 += MISO_PHILOX_W0;
    k1 += MISO_PHILOX_W1;
  }
  out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
  return out;
}

MISO_HD miso_u32x4 miso_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  return miso_philox4x32_r(MISO_PHILOX_ROUNDS, c0, c1, c2, c3, k0, k1);
}

/* The addressed form used everywhere: one block of four words. */
MISO_HD miso_u32x4 miso_draw_block(uint64_t seed, uint32_t event_id, uint32_t chain,
                                   uint32_t iteration, uint32_t site, uint32_t block) {
  return miso_philox4x32(block, iteration, site | (chain << 8), event_id,
                            (uint32_t) seed, (uint32_t) (seed >> 32));
}

/* Half-word h (0..7) of a block: the lazy-low-bits uniforms of single-end two-isoform events (header comment). */
MISO_HD uint32_t miso_block_half(miso_u32x4 b, uint32_t h) {
  const uint32_t w = (h >> 1) == 0 ? b.v[0] : ((h >> 1) == 1 ? b.v[1] : ((h >> 1) == 2 ? b.v[2] : b.v[3]));
  return (w >> (16u * (h & 1u))) & 0xFFFFu;
}
/* ... and the 32-bit uniform word of the r-th drawing read of such an event */
MISO_HD uint32_t miso_split_word(uint64_t seed, uint32_t event_id, uint32_t chain, uint32_t iteration, uint32_t r) {
  const miso_u32x4 hi = miso_draw_block(seed, event_id, chain, iteration, MISO_SITE_GIBBS, r >> 3);
  const miso_u32x4 lo = miso_draw_block(seed, event_id, chain, iteration, MISO_SITE_GIBBS_LOW, r >> 3);
  return (miso_block_half(hi, r & 7u) << 16) | miso_block_half(lo, r & 7u);
}

/* Paired-end fragment scores (miso_paired.c:157-163) are summed in fixed point so that the sum does
   not depend on the order of the reads: value * 2^MISO_SFIX_BITS rounded to nearest, int32 per
   table entry (|value| < 31), MISO_SFIX_BAD = non-finite / not representable. */
#define MISO_SFIX_BITS 26
#define MISO_SFIX_SCALE 67108864.0 /* 2^26 */
#define MISO_SFIX_BAD INT32_MIN

/* u32 -> [0,1) with 32-bit resolution; exact in double. */
MISO_HD double miso_u01(uint32_t w) { return (double) w * (1.0 / 4294967296.0); }

#endif /* MISO_PHILOX_H */

// coop.hpp -- one chain on SEVERAL workgroups: the exchange between them, once per Gibbs step.
//
// A chain can use a whole workgroup through LDS (k2_body WIDE, sampler_grp WIDE, sampler_flat FLAT_WIDE); the events
// that need more -- 10^5 read pairs with twenty isoforms: 3.4 s on 256 lanes where the rest of the batch takes 1 s --
// get N workgroups.  Every workgroup carries the chain's state and runs the scalar step redundantly (same inputs,
// same routines, same bits); the read loop's lanes of all N workgroups stride over the chain's quads; the partial
// totals meet in global memory: atomic adds into one of three rotating accumulators, one barrier per Gibbs step.
//
// Progress: a cooperative workgroup waits only for the other workgroups of ITS chain; every launch of a batch hands out at
// most COOP_MAX_WGS cooperative workgroups in all (runtime.hip counts the two-isoform plans' and the gene runs' together:
// a quarter of what the device holds), puts them at the front of their grids and starts those kernels first, and no
// other workgroup ever waits for anything -- so whatever else occupies the device (the batch's other kernels, another
// batch, another process) drains and the missing members become resident.  That is an argument about schedulers, not a
// guarantee (HIP promises no dispatch order across queues or processes), hence the second line of defence: a workgroup
// that has polled `max_polls` times (COOP_MAX_POLLS ~ tens of seconds) raises the chain's abort flag, every workgroup of
// the chain leaves, and miso_batch_sync() re-runs the launch in the same process with every chain on ONE workgroup
// (results are layout-independent bit for bit) -- a batch never fails, and never hangs the device, because of this
// (the reference's workers share nothing either, misopy/miso.py:165-187).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace miso {

constexpr int COOP_MAX_WGS = 128;          // all cooperative workgroups of a launch (runtime.hip)
constexpr int COOP_MAX_N = 32;             // workgroups of one chain
constexpr uint32_t COOP_MAX_POLLS = 1u << 25;
// per chain, in coop_mem (dwords): [0] arrivals, [1] abort, [2..3] pad, then 3 accumulators of COOP_ACC dwords:
// [0..1] score sum (int64), [2] bad flag, [3] spare, [4 ..] per-isoform totals
constexpr int COOP_ACC = 4 + 32;
constexpr int COOP_WORDS = 4 + 3 * COOP_ACC;

struct CoopGroup {
  int rank, n;          // this workgroup among the chain's n
  uint32_t *mem;        // the chain's COOP_WORDS dwords
  uint32_t max_polls = COOP_MAX_POLLS;   // (KernelArgs::coop_max_polls overrides: tests)
};

__device__ __forceinline__ uint32_t coop_load(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every workgroup of the group calls this once per step (step = 0, 1, 2 ...) after its atomic adds into accumulator
// step % 3.  Returns false when the group gave up (abort flag).
__device__ __forceinline__ bool coop_barrier(const CoopGroup &g, uint32_t step, int *lds_flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(&g.mem[0], 1u);
    const uint32_t want = (step + 1u) * static_cast<uint32_t>(g.n);
    uint32_t polls = 0;
    int ok = 1;
    while (coop_load(&g.mem[0]) < want) {
      __builtin_amdgcn_s_sleep(8);
      if (coop_load(&g.mem[1]) != 0u || ++polls > g.max_polls) { atomicExch(&g.mem[1], 1u); ok = 0; break; }
    }
    if (coop_load(&g.mem[1]) != 0u) ok = 0;
    __threadfence();
    *lds_flag = ok;
  }
  __syncthreads();
  return *lds_flag != 0;
}

__device__ __forceinline__ uint32_t *coop_acc(const CoopGroup &g, uint32_t step) { return g.mem + 4 + (step % 3u) * COOP_ACC; }

}  // namespace miso

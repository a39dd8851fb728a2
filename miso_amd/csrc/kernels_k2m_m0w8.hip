// sampler_k2_multi<0, 8> (kernels_k2m.inl)
// The one-round layout (two wavefronts per SIMD by construction, 158 of 256 registers used): the Metropolis-Hastings step's
// exp / log coefficients live in VGPRs instead of being read from constant memory at every call (detmath_n.hpp) -- a
// wavefront whose partner has finished waits out every scalar load alone.  Same operations, same bits; same-box A/B,
// round 5: 68.15 -> 67.54 ms on the headline batch, 57.8 -> 57.0 ms on hg19-like read counts; the several-rounds layouts
// (<0, 4>: three wavefronts per SIMD) lose their third wavefront to the 48 registers and stay as they were
// (profiles/r05_k2_table_registers.txt).
#define MISO_K2_TAB_REGS 1
#include "kernels_k2m.inl"
namespace miso {
template __global__ void sampler_k2_multi<0, 8>(const KernelArgs);
}

// sampler_flat for single-end events of the isoform-count class K <= 4 (see kernels_flat.inl)
#include "kernels_flat.inl"

namespace miso {
template __global__ void sampler_flat<4, 0>(const KernelArgs);   // the slice layout at run time (fallback)
template __global__ void sampler_flat<4, 3>(const KernelArgs);   // ... of 3 isoforms at compile time
template __global__ void sampler_flat<4, 4>(const KernelArgs);   // ... of 4 isoforms at compile time
template __global__ void sampler_flat<4, 3, true>(const KernelArgs);   // ... and every event of the launch has 3
template __global__ void sampler_flat<4, 4, true>(const KernelArgs);   // ... and every event of the launch has 4
}  // namespace miso

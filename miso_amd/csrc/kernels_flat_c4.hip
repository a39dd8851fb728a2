// sampler_flat for single-end events of the isoform-count class K <= 4 (see kernels_flat.inl)
#include "kernels_flat.inl"

namespace miso {
template __global__ void sampler_flat<4, 0>(const KernelArgs);   // the slice layout at run time (fallback)
template __global__ void sampler_flat<4, 3>(const KernelArgs);   // ... of 3 isoforms at compile time
template __global__ void sampler_flat<4, 4>(const KernelArgs);   // ... of 4 isoforms at compile time
}  // namespace miso

// sampler_flat for single-end events of the isoform-count class K <= 12 (see kernels_flat.inl)
#include "kernels_flat.inl"

namespace miso {
template __global__ void sampler_flat<12, 0>(const KernelArgs);   // the slice layout at run time (fallback)
template __global__ void sampler_flat<12, 9>(const KernelArgs);   // ... of 9 isoforms at compile time
template __global__ void sampler_flat<12, 10>(const KernelArgs);   // ... of 10 isoforms at compile time
template __global__ void sampler_flat<12, 11>(const KernelArgs);   // ... of 11 isoforms at compile time
template __global__ void sampler_flat<12, 12>(const KernelArgs);   // ... of 12 isoforms at compile time
template __global__ void sampler_flat<12, 9, true>(const KernelArgs);   // ... and every event of the launch has 9
template __global__ void sampler_flat<12, 10, true>(const KernelArgs);   // ... and every event of the launch has 10
template __global__ void sampler_flat<12, 11, true>(const KernelArgs);   // ... and every event of the launch has 11
template __global__ void sampler_flat<12, 12, true>(const KernelArgs);   // ... and every event of the launch has 12
}  // namespace miso

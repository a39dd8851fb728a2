// sampler_flat for single-end events of the isoform-count class K <= 8 (see kernels_flat.inl)
#include "kernels_flat.inl"

namespace miso {
template __global__ void sampler_flat<8, 0>(const KernelArgs);   // the slice layout at run time (fallback)
template __global__ void sampler_flat<8, 5>(const KernelArgs);   // ... of 5 isoforms at compile time
template __global__ void sampler_flat<8, 6>(const KernelArgs);   // ... of 6 isoforms at compile time
template __global__ void sampler_flat<8, 7>(const KernelArgs);   // ... of 7 isoforms at compile time
template __global__ void sampler_flat<8, 8>(const KernelArgs);   // ... of 8 isoforms at compile time
template __global__ void sampler_flat<8, 5, true>(const KernelArgs);   // ... and every event of the launch has 5
template __global__ void sampler_flat<8, 6, true>(const KernelArgs);   // ... and every event of the launch has 6
template __global__ void sampler_flat<8, 7, true>(const KernelArgs);   // ... and every event of the launch has 7
template __global__ void sampler_flat<8, 8, true>(const KernelArgs);   // ... and every event of the launch has 8
}  // namespace miso

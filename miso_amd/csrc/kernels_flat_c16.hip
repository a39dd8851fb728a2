// sampler_flat for single-end events of the isoform-count class K <= 16 (see kernels_flat.inl)
#include "kernels_flat.inl"

namespace miso {
template __global__ void sampler_flat<16, 0>(const KernelArgs);   // the slice layout at run time
template __global__ void sampler_flat<16, 13>(const KernelArgs);   // ... of 13 isoforms at compile time
template __global__ void sampler_flat<16, 14>(const KernelArgs);   // ... of 14 isoforms at compile time
template __global__ void sampler_flat<16, 15>(const KernelArgs);   // ... of 15 isoforms at compile time
template __global__ void sampler_flat<16, 16>(const KernelArgs);   // ... of 16 isoforms at compile time
template __global__ void sampler_flat<16, 13, true>(const KernelArgs);   // ... and every event of the launch has 13
template __global__ void sampler_flat<16, 14, true>(const KernelArgs);   // ... and every event of the launch has 14
template __global__ void sampler_flat<16, 15, true>(const KernelArgs);   // ... and every event of the launch has 15
template __global__ void sampler_flat<16, 16, true>(const KernelArgs);   // ... and every event of the launch has 16
}  // namespace miso

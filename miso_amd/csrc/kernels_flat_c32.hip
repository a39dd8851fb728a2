// sampler_flat for single-end events of the isoform-count class K <= 32 (see kernels_flat.inl)
#include "kernels_flat.inl"

namespace miso {
template __global__ void sampler_flat<32, 0>(const KernelArgs);   // the slice layout at run time
template __global__ void sampler_flat<32, 17>(const KernelArgs);   // ... of 17 isoforms at compile time
template __global__ void sampler_flat<32, 18>(const KernelArgs);   // ... of 18 isoforms at compile time
template __global__ void sampler_flat<32, 19>(const KernelArgs);   // ... of 19 isoforms at compile time
template __global__ void sampler_flat<32, 20>(const KernelArgs);   // ... of 20 isoforms at compile time
template __global__ void sampler_flat<32, 17, true>(const KernelArgs);   // ... and every event of the launch has 17
template __global__ void sampler_flat<32, 18, true>(const KernelArgs);   // ... and every event of the launch has 18
template __global__ void sampler_flat<32, 19, true>(const KernelArgs);   // ... and every event of the launch has 19
template __global__ void sampler_flat<32, 20, true>(const KernelArgs);   // ... and every event of the launch has 20
}  // namespace miso

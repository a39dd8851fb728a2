// sampler_pel for paired-end genes of the isoform-count class K <= 32 (see kernels_pel.inl)
#include "kernels_pel.inl"

namespace miso {
template __global__ void sampler_pel<16, 32>(const KernelArgs);
template __global__ void sampler_pel<32, 32>(const KernelArgs);
}  // namespace miso

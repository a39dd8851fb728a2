// sampler_pel for paired-end genes of the isoform-count class K <= 8 (see kernels_pel.inl)
#include "kernels_pel.inl"

namespace miso {
template __global__ void sampler_pel<16, 8>(const KernelArgs);
template __global__ void sampler_pel<32, 8>(const KernelArgs);
}  // namespace miso

// sampler_pel for paired-end genes of the isoform-count class K <= 4 (see kernels_pel.inl)
#include "kernels_pel.inl"

namespace miso {
template __global__ void sampler_pel<16, 4>(const KernelArgs);
template __global__ void sampler_pel<32, 4>(const KernelArgs);
}  // namespace miso

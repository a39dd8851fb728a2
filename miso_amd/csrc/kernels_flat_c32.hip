// sampler_flat for single-end events of the isoform-count class K <= 32 (see kernels_flat.inl)
#include "kernels_flat.inl"

namespace miso {
template __global__ void sampler_flat<32, 0>(const KernelArgs);
}  // namespace miso

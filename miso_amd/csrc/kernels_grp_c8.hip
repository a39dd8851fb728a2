// sampler_grp for events with 5-8 isoforms (see kernels_grp.inl)
#include "kernels_grp.inl"

namespace miso {
#define MISO_INSTANTIATE_GRP(G) \
  template __global__ void sampler_grp<G, false, 8>(const KernelArgs); \
  template __global__ void sampler_grp<G, true, 8>(const KernelArgs);
MISO_INSTANTIATE_GRP(2)
MISO_INSTANTIATE_GRP(4)
MISO_INSTANTIATE_GRP(8)
MISO_INSTANTIATE_GRP(16)
MISO_INSTANTIATE_GRP(32)
template __global__ void sampler_grp<64, true, 8, true>(const KernelArgs);   // one chain per workgroup
template __global__ void sampler_grp<64, true, 8>(const KernelArgs);         // one chain per wavefront
template __global__ void sampler_grp_multi<8>(const KernelArgs);             // the class's size buckets in one launch
}  // namespace miso

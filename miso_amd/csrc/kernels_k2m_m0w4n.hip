// sampler_k2_multi<0, 4, true> (kernels_k2m.inl): the several-rounds layout for plans of one and two lanes per chain only,
// three wavefronts per SIMD
#include "kernels_k2m.inl"
namespace miso {
template __global__ void sampler_k2_multi<0, 4, true>(const KernelArgs);
}

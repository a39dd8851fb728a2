// sampler_k2_multi<1, 4> (kernels_k2m.inl)
#include "kernels_k2m.inl"
namespace miso {
template __global__ void sampler_k2_multi<1, 4>(const KernelArgs);
}

// kernels_k2.hip -- the single-width launches of the two-isoform sampler (kernels_k2.inl) and the launch with
// two lane widths; the launch with a lane width per event lives in kernels_k2m.hip.
#include "kernels_k2.inl"

namespace miso {

template <int G, int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void sampler_k2(const KernelArgs a) {
  k2_body<G, MODE, WPB>(a, blockIdx.x, gridDim.x);
}

// Single-end, one launch, two lane widths: the events with the most drawing reads (the first mix_slots of the
// launch's list, which is ordered by them) get GA = GB + 1 lanes per chain, the rest GB.  The single-width
// launch leaves wave slots empty whenever chains / (64 / G) is not the device's slot count (40 000 chains at
// 21 per wavefront: 1905 of 2048) and its heaviest wavefront pair sets the kernel's duration; the split fills
// every CU with one 8-wavefront workgroup and gives the heavy events shorter loops.  Two separate launches do
// not work: the dispatcher then sometimes gives a CU to both kernels in turn (101 ms or 182 ms per launch,
// at random).
template <int GA, int GB>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void sampler_k2_mix(const KernelArgs a) {
  KernelArgs part = a;
  if (blockIdx.x < static_cast<unsigned>(a.mix_blocks)) {
    part.n_slots = a.mix_slots;
    k2_body<GA, 0, 8>(part, blockIdx.x, static_cast<unsigned>(a.mix_blocks));
  } else {
    part.slot_event = a.slot_event + a.mix_slots;
    part.n_slots = a.n_slots - a.mix_slots;
    k2_body<GB, 0, 8>(part, blockIdx.x - static_cast<unsigned>(a.mix_blocks), gridDim.x - static_cast<unsigned>(a.mix_blocks));
  }
}
template __global__ void sampler_k2_mix<2, 1>(const KernelArgs);
template __global__ void sampler_k2_mix<3, 2>(const KernelArgs);
template __global__ void sampler_k2_mix<4, 3>(const KernelArgs);
template __global__ void sampler_k2_mix<5, 4>(const KernelArgs);
template __global__ void sampler_k2_mix<6, 5>(const KernelArgs);
template __global__ void sampler_k2_mix<7, 6>(const KernelArgs);
template __global__ void sampler_k2_mix<8, 7>(const KernelArgs);

#define MISO_INSTANTIATE_K2(G)                                     \
  template __global__ void sampler_k2<G, 0, 4>(const KernelArgs); \
  template __global__ void sampler_k2<G, 0, 8>(const KernelArgs); \
  template __global__ void sampler_k2<G, 1, 4>(const KernelArgs); \
  template __global__ void sampler_k2<G, 2, 4>(const KernelArgs);
MISO_INSTANTIATE_K2(1)
MISO_INSTANTIATE_K2(2)
MISO_INSTANTIATE_K2(3)
MISO_INSTANTIATE_K2(4)
MISO_INSTANTIATE_K2(5)
MISO_INSTANTIATE_K2(6)
MISO_INSTANTIATE_K2(7)
MISO_INSTANTIATE_K2(8)
MISO_INSTANTIATE_K2(9)
MISO_INSTANTIATE_K2(10)
MISO_INSTANTIATE_K2(12)
MISO_INSTANTIATE_K2(16)
MISO_INSTANTIATE_K2(21)
MISO_INSTANTIATE_K2(32)
MISO_INSTANTIATE_K2(64)

}  // namespace miso

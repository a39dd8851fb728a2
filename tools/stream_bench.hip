// Micro-benchmark: what the vector memory path (TA / vector L1) charges a wavefront for streaming small
// per-chain records out of L2, by access shape -- the paired-end read loops fetch 8..168 bytes per lane and trip.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o tools/bin/stream_bench
// 8 wavefronts per CU, every wavefront walks its own 16 KB region (128 KB per CU: beyond the 32 KB L1, inside
// the 4 MB L2 of its XCD), batches of 4 loads, nothing else in the loop.  Reported: CU cycles per wave64 load
// instruction and bytes per CU cycle (wall clock x 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int REGION = 16 * 1024;

template <int BYTES>
__global__ __launch_bounds__(512) void stream(uint32_t *out, const unsigned char *buf, int n, int lane_stride, int group, int group_stride, int step) {
  const int lane = threadIdx.x & 63;
  const long wave = (static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  // lanes in groups of `group`: within a group lane_stride bytes apart, groups group_stride bytes apart
  const uint32_t lane_off = static_cast<uint32_t>((lane % group) * lane_stride + (lane / group) * group_stride);
  const unsigned char *base = buf + static_cast<size_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(wave))) * REGION;
  uint32_t acc = 0, pos = 0;
  for (int it = 0; it < n; it++) {
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { o[i] = (pos + lane_off) & (REGION - 1); pos += step; }
    if constexpr (BYTES == 4) {
      uint32_t v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) asm volatile("global_load_dword %0, %1, %2" : "=v"(v[i]) : "v"(o[i]), "s"(base));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; i++) acc ^= v[i];
    } else {
      uint4 v[4];
#pragma unroll
      for (int i = 0; i < 4; i++) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[i]) : "v"(o[i]), "s"(base));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; i++) acc ^= v[i].x ^ v[i].w;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, waves = 8, n = 3000;
  const size_t bytes = static_cast<size_t>(cus) * waves * REGION;
  unsigned char *buf; uint32_t *out;
  HIP_OK(hipMalloc(&buf, bytes)); HIP_OK(hipMemset(buf, 1, bytes));
  HIP_OK(hipMalloc(&out, static_cast<size_t>(cus) * waves * 64 * 4));
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  struct Case { const char *name; int bytes, lane_stride, group, group_stride, step; };
  const Case cases[] = {
      {"dword   fully coalesced (64 lanes x 4 B)", 4, 4, 64, 0, 256},
      {"dword   16 lanes x 4 B, 4 groups 1 KB apart", 4, 4, 16, 1024, 64},
      {"dword   8 lanes x 4 B, 8 groups 1 KB apart", 4, 4, 8, 1024, 32},
      {"dword   every lane its own 128 B line", 4, 128, 64, 0, 4},
      {"dword   every lane 48 B apart", 4, 48, 64, 0, 4},
      {"dwordx4 fully coalesced (64 lanes x 16 B)", 16, 16, 64, 0, 1024},
      {"dwordx4 16 lanes x 16 B, 4 groups 1 KB apart", 16, 16, 16, 1024, 256},
      {"dwordx4 every lane 48 B apart (K=5 indices)", 16, 48, 64, 0, 16},
      {"dwordx4 every lane 128 B apart (K=5 + scores)", 16, 128, 64, 0, 16},
      {"dwordx4 16 lanes 48 B apart, 4 groups 4 KB apart", 16, 48, 16, 4096, 16},
  };
  printf("device %s  CUs %d, %d waves per CU, %d KB per wave\n", prop.name, cus, waves, REGION / 1024);
  printf("%-52s %16s %16s\n", "access shape", "cycles / instr", "bytes / cycle");
  for (const Case &c : cases) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      HIP_OK(hipEventRecord(e0, nullptr));
      if (c.bytes == 4) hipLaunchKernelGGL(stream<4>, dim3(cus), dim3(64 * waves), 0, 0, out, buf, n, c.lane_stride, c.group, c.group_stride, c.step);
      else hipLaunchKernelGGL(stream<16>, dim3(cus), dim3(64 * waves), 0, 0, out, buf, n, c.lane_stride, c.group, c.group_stride, c.step);
      HIP_OK(hipEventRecord(e1, nullptr)); HIP_OK(hipEventSynchronize(e1));
      float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double cyc = best * 1e-3 * 2.4e9 / (static_cast<double>(waves) * n * 4);
    printf("%-52s %16.2f %16.1f\n", c.name, cyc, 64.0 * c.bytes / cyc);
  }
  return 0;
}

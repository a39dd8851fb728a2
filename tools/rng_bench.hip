// Micro-benchmark: cost per 32-bit output of counter-based generators on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/rng_bench.hip -o /tmp/rng_bench && /tmp/rng_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

template <int R> __device__ __forceinline__ void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *o) {
#pragma unroll
  for (int r = 0; r < R; r++) {
    uint64_t p0 = (uint64_t) 0xD2511F53u * c0, p1 = (uint64_t) 0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t) (p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t) (p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t) p1; c3 = (uint32_t) p0; c0 = n0; c2 = n2; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
template <int R> __device__ __forceinline__ void threefry(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t *o) {
  const int R0[8] = {10, 11, 13, 23, 6, 17, 25, 18}, R1[8] = {26, 21, 27, 5, 20, 11, 10, 20};
  uint32_t ks[5] = {k0, k1, k2, k3, 0x1BD11BDAu ^ k0 ^ k1 ^ k2 ^ k3};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1], x2 = c2 + ks[2], x3 = c3 + ks[3];
#pragma unroll
  for (int r = 0; r < R; r++) {
    if ((r & 1) == 0) { x0 += x1; x1 = rotl(x1, R0[r & 7]) ^ x0; x2 += x3; x3 = rotl(x3, R1[r & 7]) ^ x2; }
    else { x0 += x3; x3 = rotl(x3, R0[r & 7]) ^ x0; x2 += x1; x1 = rotl(x1, R1[r & 7]) ^ x2; }
    if ((r & 3) == 3) { const int s = (r + 1) >> 2; x0 += ks[s % 5]; x1 += ks[(s + 1) % 5]; x2 += ks[(s + 2) % 5]; x3 += ks[(s + 3) % 5] + s; }
  }
  o[0] = x0; o[1] = x1; o[2] = x2; o[3] = x3;
}

template <int WHICH> __global__ void bench(uint32_t *out, int n, uint32_t seed) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0, o[4];
  for (int i = 0; i < n; i += 2) {  // two independent blocks per trip (ILP 2)
    uint32_t o2[4];
    if (WHICH == 0) { philox<10>(i, tid, 2, 7, seed, 1, o); philox<10>(i + 1, tid, 2, 7, seed, 1, o2); }
    if (WHICH == 1) { philox<7>(i, tid, 2, 7, seed, 1, o); philox<7>(i + 1, tid, 2, 7, seed, 1, o2); }
    if (WHICH == 2) { threefry<20>(i, tid, 0, 0, seed, 1, 7, 2, o); threefry<20>(i + 1, tid, 0, 0, seed, 1, 7, 2, o2); }
    if (WHICH == 3) { threefry<12>(i, tid, 0, 0, seed, 1, 7, 2, o); threefry<12>(i + 1, tid, 0, 0, seed, 1, 7, 2, o2); }
    acc += (o[0] < seed) + (o[1] < seed) + (o[2] < seed) + (o[3] < seed) + (o2[0] < seed) + (o2[1] < seed) + (o2[2] < seed) + (o2[3] < seed);
  }
  out[tid] = acc;
}

__global__ void kat(uint32_t *out) {
  threefry<20>(0, 0, 0, 0, 0, 0, 0, 0, out);
  threefry<20>(~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, out + 4);
  threefry<20>(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0, 0x082efa98, 0xec4e6c89, out + 8);
  philox<10>(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0, out + 12);
}

int main() {
  uint32_t *d; hipMalloc(&d, 1 << 24);
  uint32_t h[16];
  hipLaunchKernelGGL(kat, dim3(1), dim3(1), 0, 0, d); hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; i++) printf("%08x%s", h[i], (i & 3) == 3 ? "\n" : " ");
  const int blocks = 256 * 8 * 4, n = 4096;  // 8 waves/SIMD
  const char *names[4] = {"philox4x32-10", "philox4x32-7", "threefry4x32-20", "threefry4x32-12"};
  for (int w = 0; w < 4; w++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (w == 0) hipLaunchKernelGGL(bench<0>, dim3(blocks), dim3(256), 0, 0, d, n, 12345u);
      if (w == 1) hipLaunchKernelGGL(bench<1>, dim3(blocks), dim3(256), 0, 0, d, n, 12345u);
      if (w == 2) hipLaunchKernelGGL(bench<2>, dim3(blocks), dim3(256), 0, 0, d, n, 12345u);
      if (w == 3) hipLaunchKernelGGL(bench<3>, dim3(blocks), dim3(256), 0, 0, d, n, 12345u);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double outputs = (double) blocks * 256 * n * 4;
    printf("%-18s %8.2f ms  %7.1f G outputs/s  (%.3f SIMD-cycles/output at 2.4 GHz)\n", names[w], ms, outputs / ms / 1e6,
           ms * 1e-3 * 1024 * 2.4e9 / outputs);
  }
  return 0;
}

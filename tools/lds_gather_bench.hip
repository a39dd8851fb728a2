// Micro-benchmark: what a table gather costs the CU on gfx950 -- the paired-end read loops do 2..K+1
// of them per read (fragment-length probability, 8 bytes; fixed-point score, 4 bytes).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_gather_bench.hip -o /tmp/lds_gather_bench && /tmp/lds_gather_bench
// One workgroup of W wavefronts per CU; every lane holds 16 table offsets in registers and issues
// ds_read / global_load batches of 8 with nothing else in the loop but one xor per load, so the LDS
// (or the vector L1) is the only busy unit.  Reported: CU cycles per wave64 gather instruction
// (wall clock x 2.4 GHz / instructions per CU) -- the LDS pipe is shared by the CU's four SIMDs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Mode { B32_RANDOM, B64_RANDOM, B64_REP32, B64_REP16, B64_REP8, B64_REP4, B128_RANDOM, B64_BROADCAST, B64_GROUP8, B64_LINEAR,
            B32_LINEAR, G64_RANDOM, G32_RANDOM, N_MODES };
static const char *mode_name[N_MODES] = {
    "ds_read_b32  random (482 entries)", "ds_read_b64  random (241 entries)",
    "ds_read_b64  32 copies, lane's own banks", "ds_read_b64  16 copies", "ds_read_b64  8 copies", "ds_read_b64  4 copies",
    "ds_read_b128 random (241 entries)", "ds_read_b64  one address per wave", "ds_read_b64  one address per 8 lanes",
    "ds_read_b64  consecutive lanes", "ds_read_b32  consecutive lanes",
    "global_load_dwordx2 random (2 KB table, L1)", "global_load_dword random (2 KB table, L1)"};

constexpr int IL = 241;

template <int BYTES, bool GLOBAL>
__global__ __launch_bounds__(1024) void gather(uint32_t *out, int n, const uint32_t *offs, const unsigned char *gtab, int tab_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x * 4; i < tab_bytes; i += blockDim.x * 4)
    *reinterpret_cast<uint32_t *>(lds + i) = *reinterpret_cast<const uint32_t *>(gtab + i);
  __syncthreads();
  uint32_t o[16];
  for (int i = 0; i < 16; i++) o[i] = offs[(blockIdx.x * blockDim.x + threadIdx.x) % (64 * 1024) * 16 + i];
  uint32_t acc = 0;
  for (int it = 0; it < n; it++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if constexpr (!GLOBAL) {
        if constexpr (BYTES == 4) {
          uint32_t v[8];
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(o[8 * h + i]));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(v[i]));
        } else if constexpr (BYTES == 8) {
          uint64_t v[8];
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("ds_read_b64 %0, %1" : "=v"(v[i]) : "v"(o[8 * h + i]));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(static_cast<uint32_t>(v[i])));
        } else {
          uint4 v[8];
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(o[8 * h + i]));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(v[i].x));
        }
      } else {
        if constexpr (BYTES == 4) {
          uint32_t v[8];
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("global_load_dword %0, %1, %2" : "=v"(v[i]) : "v"(o[8 * h + i]), "s"(gtab));
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(v[i]));
        } else {
          uint64_t v[8];
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v[i]) : "v"(o[8 * h + i]), "s"(gtab));
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 8; i++) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(static_cast<uint32_t>(v[i])));
        }
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d\n", prop.name, cus);
  const int n = 4000, nthreads_tab = 64 * 1024;
  uint32_t *d_out, *d_offs; unsigned char *d_tab;
  HIP_OK(hipMalloc(&d_out, static_cast<size_t>(cus) * 1024 * 4));
  HIP_OK(hipMalloc(&d_offs, static_cast<size_t>(nthreads_tab) * 16 * 4));
  HIP_OK(hipMalloc(&d_tab, 64 * 1024));
  HIP_OK(hipMemset(d_tab, 1, 64 * 1024));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  printf("%-46s %12s %12s %12s\n", "gather (CU cycles per wave64 instruction)", "4 waves/CU", "8 waves/CU", "16 waves/CU");
  for (int m = 0; m < N_MODES; m++) {
    std::vector<uint32_t> offs(static_cast<size_t>(nthreads_tab) * 16);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return static_cast<uint32_t>(s >> 11); };
    int bytes = 8, tab_bytes = IL * 8; bool glob = false;
    for (int t = 0; t < nthreads_tab; t++) {
      const int lane = t & 63;
      for (int i = 0; i < 16; i++) {
        uint32_t &o = offs[static_cast<size_t>(t) * 16 + i];
        const uint32_t f = rnd() % IL;
        switch (m) {
        case B32_RANDOM: bytes = 4; tab_bytes = 2 * IL * 4; o = (rnd() % (2 * IL)) * 4; break;
        case B64_RANDOM: o = f * 8; break;
        case B64_REP32: tab_bytes = IL * 32 * 8; o = (f * 32 + (lane & 31)) * 8; break;
        case B64_REP16: tab_bytes = IL * 16 * 8; o = (f * 16 + (lane & 15)) * 8; break;
        case B64_REP8: tab_bytes = IL * 8 * 8; o = (f * 8 + (lane & 7)) * 8; break;
        case B64_REP4: tab_bytes = IL * 4 * 8; o = (f * 4 + (lane & 3)) * 8; break;
        case B128_RANDOM: bytes = 16; tab_bytes = IL * 16; o = f * 16; break;
        case B64_BROADCAST: o = ((i * 37 + 11) % IL) * 8; break;
        case B64_GROUP8: o = (((lane >> 3) * 29 + i * 37) % IL) * 8; break;
        case B64_LINEAR: o = ((lane + i * 7) % IL) * 8; break;
        case B32_LINEAR: bytes = 4; tab_bytes = 2 * IL * 4; o = ((lane + i * 7) % (2 * IL)) * 4; break;
        case G64_RANDOM: glob = true; o = f * 8; break;
        case G32_RANDOM: glob = true; bytes = 4; tab_bytes = 2 * IL * 4; o = (rnd() % (2 * IL)) * 4; break;
        }
      }
    }
    HIP_OK(hipMemcpy(d_offs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice));
    printf("%-46s", mode_name[m]);
    for (int waves : {4, 8, 16}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        HIP_OK(hipEventRecord(e0, nullptr));
        const dim3 g(cus), b(64 * waves);
        const size_t lds = 64 * 1024;
        if (glob && bytes == 8) hipLaunchKernelGGL((gather<8, true>), g, b, lds, 0, d_out, n, d_offs, d_tab, tab_bytes);
        else if (glob) hipLaunchKernelGGL((gather<4, true>), g, b, lds, 0, d_out, n, d_offs, d_tab, tab_bytes);
        else if (bytes == 4) hipLaunchKernelGGL((gather<4, false>), g, b, lds, 0, d_out, n, d_offs, d_tab, tab_bytes);
        else if (bytes == 8) hipLaunchKernelGGL((gather<8, false>), g, b, lds, 0, d_out, n, d_offs, d_tab, tab_bytes);
        else hipLaunchKernelGGL((gather<16, false>), g, b, lds, 0, d_out, n, d_offs, d_tab, tab_bytes);
        HIP_OK(hipEventRecord(e1, nullptr));
        HIP_OK(hipEventSynchronize(e1));
        float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf(" %12.2f", best * 1e-3 * 2.4e9 / (static_cast<double>(waves) * n * 16));
    }
    printf("\n");
  }
  return 0;
}

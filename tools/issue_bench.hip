// Micro-benchmark: VALU issue cost (SIMD cycles per wave64 instruction) of the instruction classes the
// sampler kernels are made of, on gfx950.  The table it prints is what bench.py's `roofline` (bound
// "valu") prices the kernels' instruction streams with (profiles/r02_issue_costs.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_bench.hip -o /tmp/issue_bench && /tmp/issue_bench
// Method: W wavefronts per SIMD (W = 1, 2, 4), each running 8 independent dependency chains of the
// instruction under test, 64 instructions per loop trip, s_memtime around the loop (shader cycles),
// and the whole launch timed with HIP events (wall clock -> effective issue rate of the chip).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// ---- one kernel per instruction: r0..r7 are independent chains ----
#define DEF_KERNEL_32(NAME, ASM)                                                             \
  __global__ void NAME(uint32_t *out, int n, uint32_t seed, unsigned long long *cyc) {       \
    uint32_t r0 = threadIdx.x + seed, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3,     \
             r4 = r0 * 11 + 4, r5 = r0 * 13 + 5, r6 = r0 * 17 + 6, r7 = r0 * 19 + 7;         \
    const uint32_t a = seed | 1u, b = seed * 0x9E3779B9u;                                    \
    const unsigned long long t0 = __builtin_readcyclecounter();                              \
    for (int i = 0; i < n; i++) {                                                            \
      _Pragma("unroll") for (int u = 0; u < 8; u++) {                                        \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                 \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                     : "v"(a), "v"(b));                                                      \
      }                                                                                      \
    }                                                                                        \
    const unsigned long long t1 = __builtin_readcyclecounter();                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;      \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                 \
  }

#define DEF_KERNEL_64(NAME, ASM)                                                             \
  __global__ void NAME(uint32_t *out, int n, uint32_t seed, unsigned long long *cyc) {       \
    double r0 = 1.0 + threadIdx.x * 1e-3 + seed * 1e-9, r1 = r0 + 0.1, r2 = r0 + 0.2, r3 = r0 + 0.3, \
           r4 = r0 + 0.4, r5 = r0 + 0.5, r6 = r0 + 0.6, r7 = r0 + 0.7;                       \
    const double a = 1.0 + seed * 1e-12, b = 1e-9 * seed;                                    \
    const unsigned long long t0 = __builtin_readcyclecounter();                              \
    for (int i = 0; i < n; i++) {                                                            \
      _Pragma("unroll") for (int u = 0; u < 8; u++) {                                        \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                 \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                     : "v"(a), "v"(b), "v"(seed));                                           \
      }                                                                                      \
    }                                                                                        \
    const unsigned long long t1 = __builtin_readcyclecounter();                              \
    const double s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                  \
    uint64_t bits; memcpy(&bits, &s, 8);                                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t) bits ^ (uint32_t) (bits >> 32);  \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                                 \
  }

// operands: %0..%7 chains, %8 = a, %9 = b
#define A_ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_BITOP3(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define A_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CMP_U32(i) "v_cmp_le_u32 vcc, %" #i ", %8\n"
#define A_CMP_ADDC(i) "v_cmp_le_u32 vcc, %" #i ", %8\n v_addc_co_u32 %" #i ", vcc, 0, %" #i ", vcc\n"
#define A_MUL_LO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_MUL_HI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define A_MAD_U32_U24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_LSHL_ADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define A_MOV_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_CVT_F64_U32(i) "v_cvt_f64_u32 %" #i ", %10\n"   /* 64-bit dst, 32-bit src */
#define A_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %9\n"
#define A_MUL_F64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define A_FMA_F64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define A_RCP_F64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define A_CMP_F64(i) "v_cmp_lt_f64 vcc, %" #i ", %8\n"
#define A_LDEXP_F64(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define A_DIV_SCALE_F64(i) "v_div_scale_f64 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define A_DIV_FMAS_F64(i) "v_div_fmas_f64 %" #i ", %" #i ", %8, %9\n"
#define A_DIV_FIXUP_F64(i) "v_div_fixup_f64 %" #i ", %" #i ", %8, %9\n"
#define A_CVT_U32_F64(i) "v_cvt_u32_f64 %" #i ", %" #i "\n"
#define A_FLOOR_F64(i) "v_floor_f64 %" #i ", %" #i "\n"

DEF_KERNEL_32(k_add_u32, A_ADD_U32)
DEF_KERNEL_32(k_xor, A_XOR)
DEF_KERNEL_32(k_bitop3, A_BITOP3)
DEF_KERNEL_32(k_cndmask, A_CNDMASK)
DEF_KERNEL_32(k_cmp_u32, A_CMP_U32)
DEF_KERNEL_32(k_cmp_addc, A_CMP_ADDC)
DEF_KERNEL_32(k_mul_lo, A_MUL_LO)
DEF_KERNEL_32(k_mul_hi, A_MUL_HI)
DEF_KERNEL_32(k_mad_u32_u24, A_MAD_U32_U24)
DEF_KERNEL_32(k_lshl_add, A_LSHL_ADD)
DEF_KERNEL_32(k_mov_dpp, A_MOV_DPP)
DEF_KERNEL_64(k_cvt_f64_u32, A_CVT_F64_U32)
DEF_KERNEL_64(k_add_f64, A_ADD_F64)
DEF_KERNEL_64(k_mul_f64, A_MUL_F64)
DEF_KERNEL_64(k_fma_f64, A_FMA_F64)
DEF_KERNEL_64(k_rcp_f64, A_RCP_F64)
DEF_KERNEL_64(k_cmp_f64, A_CMP_F64)
DEF_KERNEL_64(k_ldexp_f64, A_LDEXP_F64)
DEF_KERNEL_64(k_div_scale_f64, A_DIV_SCALE_F64)
DEF_KERNEL_64(k_div_fmas_f64, A_DIV_FMAS_F64)
DEF_KERNEL_64(k_div_fixup_f64, A_DIV_FIXUP_F64)
DEF_KERNEL_64(k_floor_f64, A_FLOOR_F64)

// v_mad_u64_u32 (Philox's multiply): eight independent destinations, fixed 32-bit sources -- throughput
__global__ void k_mad_u64_u32(uint32_t *out, int n, uint32_t seed, unsigned long long *cyc) {
  uint64_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0, r5 = 0, r6 = 0, r7 = 0;
  const uint32_t a = 0xD2511F53u, s = threadIdx.x + seed;
  const uint64_t z = seed;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %10\n v_mad_u64_u32 %1, vcc, %8, %9, %10\n v_mad_u64_u32 %2, vcc, %8, %9, %10\n"
                   "v_mad_u64_u32 %3, vcc, %8, %9, %10\n v_mad_u64_u32 %4, vcc, %8, %9, %10\n v_mad_u64_u32 %5, vcc, %8, %9, %10\n"
                   "v_mad_u64_u32 %6, vcc, %8, %9, %10\n v_mad_u64_u32 %7, vcc, %8, %9, %10\n"
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
                   : "v"(s), "v"(a), "v"(z)
                   : "vcc");
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const uint64_t x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t) x ^ (uint32_t) (x >> 32);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// LDS: ds_read_b32 / ds_read_b64 / ds_add_u32, conflict-free addresses (lane * 4 / lane * 8)
__global__ void k_ds_read_b32(uint32_t *out, int n, uint32_t seed, unsigned long long *cyc) {
  __shared__ uint32_t lds[4608];
  for (int i = threadIdx.x; i < 4608; i += blockDim.x) lds[i] = i * seed;
  __syncthreads();
  uint32_t acc = 0;
  const uint32_t addr = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 1024;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) {
    uint32_t v0, v1, v2, v3, v4, v5, v6, v7;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                   "ds_read_b32 %4, %8 offset:4\n ds_read_b32 %5, %8 offset:260\n ds_read_b32 %6, %8 offset:516\n ds_read_b32 %7, %8 offset:772\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                   : "v"(addr));
      acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k_ds_read_b64(uint32_t *out, int n, uint32_t seed, unsigned long long *cyc) {
  __shared__ uint64_t lds[4608];
  for (int i = threadIdx.x; i < 4608; i += blockDim.x) lds[i] = i * seed;
  __syncthreads();
  uint64_t acc = 0;
  const uint32_t addr = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 2048;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; i++) {
    uint64_t v0, v1, v2, v3, v4, v5, v6, v7;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n ds_read_b64 %3, %8 offset:1536\n"
                   "ds_read_b64 %4, %8 offset:8\n ds_read_b64 %5, %8 offset:520\n ds_read_b64 %6, %8 offset:1032\n ds_read_b64 %7, %8 offset:1544\n"
                   "s_waitcnt lgkmcnt(0)\n"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7)
                   : "v"(addr));
      acc ^= v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t) acc ^ (uint32_t) (acc >> 32);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

typedef void (*kern_t)(uint32_t *, int, uint32_t, unsigned long long *);
struct Entry { const char *name; kern_t k; int per_trip; };

int main() {
  uint32_t *d; unsigned long long *dc;
  hipMalloc(&d, 256u * 1024 * 4 * 2);
  hipMalloc(&dc, 8);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("device %s  CUs %d  clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  Entry tab[] = {
    {"v_add_u32", k_add_u32, 64}, {"v_xor_b32", k_xor, 64}, {"v_bitop3_b32", k_bitop3, 64},
    {"v_cndmask_b32", k_cndmask, 64}, {"v_cmp_le_u32", k_cmp_u32, 64}, {"v_cmp+v_addc pair", k_cmp_addc, 64},
    {"v_lshl_add_u32", k_lshl_add, 64}, {"v_mov_b32 dpp", k_mov_dpp, 64},
    {"v_mul_lo_u32", k_mul_lo, 64}, {"v_mul_hi_u32", k_mul_hi, 64}, {"v_mad_u32_u24", k_mad_u32_u24, 64},
    {"v_mad_u64_u32", k_mad_u64_u32, 64},
    {"v_cvt_f64_u32", k_cvt_f64_u32, 64}, {"v_add_f64", k_add_f64, 64}, {"v_mul_f64", k_mul_f64, 64},
    {"v_fma_f64", k_fma_f64, 64}, {"v_cmp_lt_f64", k_cmp_f64, 64}, {"v_ldexp_f64", k_ldexp_f64, 64},
    {"v_floor_f64", k_floor_f64, 64},
    {"v_rcp_f64", k_rcp_f64, 64}, {"v_div_scale_f64", k_div_scale_f64, 64}, {"v_div_fmas_f64", k_div_fmas_f64, 64},
    {"v_div_fixup_f64", k_div_fixup_f64, 64},
    {"ds_read_b32 (8 + wait)", k_ds_read_b32, 64}, {"ds_read_b64 (8 + wait)", k_ds_read_b64, 64},
  };
  const int n = 2000;
  printf("%-24s %28s %28s %28s\n", "", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
  printf("%-24s %13s %14s %13s %14s %13s %14s\n", "instruction", "cyc/inst(1w)", "cyc/inst@2.4G", "cyc/inst(1w)", "cyc/inst@2.4G",
         "cyc/inst(1w)", "cyc/inst@2.4G");
  for (const Entry &e : tab) {
    printf("%-24s", e.name);
    for (int w : {1, 2, 4}) {
      // one block of 256 w threads per CU (w wavefronts on each of its 4 SIMDs); 90 KB of dynamic LDS
      // per block keeps a second block off the CU, so the placement is known
      const int blocks = cus;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0;
      hipFuncSetAttribute(reinterpret_cast<const void *>(e.k), hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256 * w), 90 * 1024, 0, d, n, 12345u, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      unsigned long long cyc = 0; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
      const double insts = (double) n * e.per_trip;
      // per-wave view: s_memtime ticks per instruction of ONE wave (includes the SIMD's other waves)
      // chip view: SIMD cycles at 2.4 GHz per instruction issued (wall time / instructions per SIMD)
      printf(" %13.2f %14.2f", (double) cyc / insts, ms * 1e-3 * 2.4e9 / (insts * w));
    }
    printf("\n");
  }
  return 0;
}

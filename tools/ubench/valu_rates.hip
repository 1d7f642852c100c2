// Issue cost of single VALU instructions on gfx950, one wave per SIMD: cycles per instruction over 8 independent chains.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o gpurun_out/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
#define REPS 256

#define BODY(ASM)                                                                                     \
  uint32_t x[CHAINS];                                                                                 \
  for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 2654435761u + i + seed;                        \
  uint64_t t0 = __builtin_readcyclecounter();                                                          \
  for (int r = 0; r < REPS; ++r) {                                                                    \
    _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(k));        \
  }                                                                                                   \
  uint64_t t1 = __builtin_readcyclecounter();                                                          \
  uint32_t s = 0;                                                                                     \
  for (int i = 0; i < CHAINS; ++i) s ^= x[i];                                                         \
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                    \
  if (s == 0x12345678u) out[1000] = s;

__global__ void k_mul_lo(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_mul_lo_u32 %0, %0, %1") }
__global__ void k_mul_u24(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_mul_u32_u24 %0, %0, %1") }
__global__ void k_mad_u24(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_mad_u32_u24 %0, %0, %1, %1") }
__global__ void k_add(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_add_u32 %0, %0, %1") }
__global__ void k_xor(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_xor_b32 %0, %0, %1") }
__global__ void k_alignbit(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_alignbit_b32 %0, %0, %0, 13") }
__global__ void k_xad(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_xad_u32 %0, %0, %1, %1") }
__global__ void k_exp(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_exp_f32 %0, %0") }
__global__ void k_fma(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_fma_f32 %0, %0, %1, %1") }
__global__ void k_mul_hi(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_mul_hi_u32 %0, %0, %1") }
__global__ void k_lshl_add(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_lshl_add_u32 %0, %0, 3, %1") }
__global__ void k_perm(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_perm_b32 %0, %0, %1, %1") }
__global__ void k_cmp_cnd(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_cmp_ge_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc") }
__global__ void k_cmp16hi(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_cmp_ge_u16_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:WORD_0\n v_cndmask_b32 %0, %0, %1, vcc") }
__global__ void k_bfe(uint64_t* out, uint32_t seed, uint32_t k) { BODY("v_bfe_u32 %0, %0, 16, 16") }

#define RUN(K)                                                                                \
  {                                                                                           \
    hipLaunchKernelGGL(K, dim3(256), dim3(256), 0, 0, d, 1u, 0x9E3779B1u);                     \
    hipDeviceSynchronize();                                                                   \
    uint64_t h[256];                                                                          \
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);                                        \
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];                                    \
    printf("%-12s %.2f clk per wave-instruction (1 wave/SIMD, s_memtime units)\n", #K, s / 256 / (CHAINS * REPS)); \
  }

int main() {
  uint64_t* d;
  hipMalloc(&d, 2048 * sizeof(uint64_t));
  RUN(k_add) RUN(k_xor) RUN(k_fma) RUN(k_mul_lo) RUN(k_mul_hi) RUN(k_mul_u24) RUN(k_mad_u24) RUN(k_alignbit) RUN(k_xad) RUN(k_lshl_add)
  RUN(k_perm) RUN(k_bfe) RUN(k_exp) RUN(k_cmp_cnd) RUN(k_cmp16hi)
  return 0;
}

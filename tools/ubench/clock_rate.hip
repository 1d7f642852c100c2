// What does s_memtime count?  One wave spins for a fixed number of s_memtime ticks / s_memrealtime ticks; hipEvents give wall time.
// Also: the same with all CUs busy on MFMAs (clock under matrix load).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void spin_memtime(uint64_t ticks, uint64_t* out) {
  const uint64_t t0 = __builtin_readcyclecounter();
  uint64_t t1;
  do { t1 = __builtin_readcyclecounter(); } while (t1 - t0 < ticks);
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
__global__ void spin_realtime(uint64_t ticks, uint64_t* out) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  uint64_t t1;
  do { t1 = __builtin_amdgcn_s_memrealtime(); } while (t1 - t0 < ticks);
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
// n MFMAs per wave over 8 independent accumulators (inline asm: the compiler's own loop shuffled accumulators through v_accvgpr moves)
__global__ __launch_bounds__(256) void mfma_burn(int n, uint64_t* out, float* sink) {
  f32x4 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(((threadIdx.x * 37 + i * 11) % 97) * 0.01f - 0.4f); b[i] = (__bf16)(((threadIdx.x * 13 + i * 7) % 89) * 0.01f - 0.4f); }
  const uint64_t t0 = __builtin_readcyclecounter();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[blockIdx.x] = t1 - t0; out[2048 + blockIdx.x] = r1 - r0; }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k][0];
  if (s == 123.456f) sink[0] = 1.f;
}

int main() {
  uint64_t* d; float* sink;
  hipMalloc(&d, 4096 * sizeof(uint64_t)); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(spin_memtime, dim3(1), dim3(64), 0, 0, (uint64_t)200000000ull, d); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("s_memtime:     2e8 ticks in %.3f ms -> %.1f MHz\n", ms, 2e8 / ms / 1e3);
    hipEventRecord(e0); hipLaunchKernelGGL(spin_realtime, dim3(1), dim3(64), 0, 0, (uint64_t)10000000ull, d); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("s_memrealtime: 1e7 ticks in %.3f ms -> %.1f MHz\n", ms, 1e7 / ms / 1e3);
  }
  for (int rep = 0; rep < 3; ++rep)
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int n = 800000;                                    // MFMAs per wave
    const int nb = 256 * waves_per_simd;
    hipEventRecord(e0); hipLaunchKernelGGL(mfma_burn, dim3(nb), dim3(256), 0, 0, n, d, sink); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0, rt = 0; for (int i = 0; i < nb; ++i) { avg += h[i]; rt += h[2048 + i]; } avg /= nb; rt /= nb;
    const double flops = 2.0 * 16 * 16 * 32 * (double)n * 4 * nb;
    printf("MFMA burn, %d wave(s)/SIMD on 256 CUs: %.3f ms wall, %.2f s_memtime ticks per MFMA per wave, %.1f TFLOP/s, in-kernel clock %.0f MHz\n",
           waves_per_simd, ms, avg / n, flops / ms / 1e9, avg / rt * 100.0);
  }
  return 0;
}

// Does v_pk_fma_f32 with op_sel:[0,1,0] (the LOW result reads the HIGH register of the src1 pair) return wrong results when another
// stream's MFMA kernel shares the CUs?  (Round 2's "lost update": tools/experiment_fork_wgrad.py bisected it to this instruction form in
// scalar_conv_wgrad_kernel.)  Three forms accumulate the same products over the same operands, the src1 pair coming from a ds_read2_b32
// each iteration as in that kernel:
//   A: v_pk_fma_f32 acc, g, x, acc op_sel:[0,1,0]          lo += g.lo * x.hi ; hi += g.hi * x.hi      (the suspect)
//   B: v_pk_fma_f32 acc, g, x, acc op_sel_hi:[1,0,1]        lo += g.lo * x.lo ; hi += g.hi * x.lo      (the form of the other taps)
//   C: two v_fma_f32                                         reference for A
// Each thread counts the iterations-blocks in which A differs from C (bitwise) and B from its own scalar reference; the host prints the
// totals per lane class, alone and with an MFMA burn kernel running on a second stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_burn(int n, float* sink) {
  f32x4 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(((threadIdx.x * 37 + i * 11) % 97) * 0.01f - 0.4f); b[i] = (__bf16)(((threadIdx.x * 13 + i * 7) % 89) * 0.01f - 0.4f); }
  for (int i = 0; i < n; i += 16) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k][0];
  if (s == 123.456f) sink[0] = 1.f;
}

__global__ __launch_bounds__(256) void pk_probe(int iters, unsigned* bad_a, unsigned* bad_b, const float* gsrc, float* sink) {
  __shared__ float xs[258];
  for (int i = threadIdx.x; i < 258; i += 256) xs[i] = 0.25f + 0.001f * ((i * 7 + blockIdx.x) % 61);
  __syncthreads();
  const float g0 = 0.5f + 0.01f * (threadIdx.x % 13), g1 = 0.75f + 0.01f * (threadIdx.x % 7);
  unsigned na = 0, nb = 0;
  float dummy = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x2 accA = {0.f, 0.f}, accB = {0.f, 0.f};
    float rA0 = 0.f, rA1 = 0.f, rB0 = 0.f, rB1 = 0.f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
      const int i = (k * 3 + (threadIdx.x >> 5)) & 255;
      f32x2 x, y, z;
      f32x4 gl;
      // as in the kernel: the pair the packed FMAs read has returned (counted wait), while LATER LDS reads and a global load are still in
      // flight and land in neighbouring registers during the FMAs
      asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %4 offset0:8 offset1:9\n\tds_read2_b32 %2, %4 offset0:16 offset1:17\n\t"
                   "global_load_dwordx4 %3, %5, off\n\ts_waitcnt lgkmcnt(2)"
                   : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(gl) : "v"((unsigned)(i * 4)), "v"(gsrc + (size_t)((k * 64 + threadIdx.x) & 4095) * 4) : "memory");
      const f32x2 g = {g0, g1};
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(accA) : "v"(g), "v"(x));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(accB) : "v"(g), "v"(x));
      const float x0 = x[0], x1 = x[1];
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(rA0) : "v"(g0), "v"(x1));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(rA1) : "v"(g1), "v"(x1));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(rB0) : "v"(g0), "v"(x0));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(rB1) : "v"(g1), "v"(x0));
      asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
      dummy += y[0] + y[1] + z[0] + z[1] + gl[0];
    }
    na += (__float_as_uint(accA[0]) != __float_as_uint(rA0)) + 65536u * (__float_as_uint(accA[1]) != __float_as_uint(rA1));
    nb += (__float_as_uint(accB[0]) != __float_as_uint(rB0)) + 65536u * (__float_as_uint(accB[1]) != __float_as_uint(rB1));
  }
  if (dummy == 123.456f) sink[1] = dummy;
  bad_a[blockIdx.x * 256 + threadIdx.x] = na;
  bad_b[blockIdx.x * 256 + threadIdx.x] = nb;
}

static void report(const char* tag, const std::vector<unsigned>& a, const std::vector<unsigned>& b) {
  unsigned long alo[4] = {0, 0, 0, 0}, ahi[4] = {0, 0, 0, 0}, blo = 0, bhi = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    const int pass = (i & 63) >> 4;                       // 16-lane pass of the wave
    alo[pass] += a[i] & 0xffff; ahi[pass] += a[i] >> 16; blo += b[i] & 0xffff; bhi += b[i] >> 16;
  }
  printf("%-28s form A (op_sel:[0,1,0]) mismatching blocks, LOW result by 16-lane pass: %lu %lu %lu %lu; HIGH result: %lu %lu %lu %lu; form B low/high: %lu/%lu\n",
         tag, alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3], blo, bhi);
}

int main() {
  const int blocks = 2048, iters = 400;
  unsigned *da, *db; float* sink;
  hipMalloc(&da, blocks * 256 * 4); hipMalloc(&db, blocks * 256 * 4); hipMalloc(&sink, 16); float* gsrc; hipMalloc(&gsrc, 4096 * 16 + 64); hipMemset(gsrc, 0, 4096 * 16 + 64);
  std::vector<unsigned> ha(blocks * 256), hb(blocks * 256);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(pk_probe, dim3(blocks), dim3(256), 0, s1, iters, da, db, gsrc, sink);
    hipDeviceSynchronize();
    hipMemcpy(ha.data(), da, ha.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), db, hb.size() * 4, hipMemcpyDeviceToHost);
    report("alone", ha, hb);
    hipLaunchKernelGGL(mfma_burn, dim3(1024), dim3(256), 0, s2, 400000, sink);
    hipLaunchKernelGGL(pk_probe, dim3(blocks), dim3(256), 0, s1, iters, da, db, gsrc, sink);
    hipDeviceSynchronize();
    hipMemcpy(ha.data(), da, ha.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), db, hb.size() * 4, hipMemcpyDeviceToHost);
    report("beside an MFMA kernel", ha, hb);
  }
  return 0;
}

// Shared device/host helpers for the Daft-Exprt gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define DX_OK 0
#define DX_ERR_ARG 1
#define DX_ERR_LAUNCH 2

#ifdef __cplusplus
extern "C" {
#endif
void dx_set_error(const char* fmt, ...);
void dx_prof_begin(int kernel_id, hipStream_t s);
void dx_prof_end(int kernel_id, hipStream_t s);
#ifdef __cplusplus
}
#endif

#define DX_REQUIRE(cond, ...)              \
  do {                                     \
    if (!(cond)) {                         \
      dx_set_error(__VA_ARGS__);           \
      return DX_ERR_ARG;                   \
    }                                      \
  } while (0)

#define DX_LAUNCH_CHECK(name)                                         \
  do {                                                                \
    hipError_t e_ = hipGetLastError();                                \
    if (e_ != hipSuccess) {                                           \
      dx_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return DX_ERR_LAUNCH;                                           \
    }                                                                 \
  } while (0)

// kernel families that can be bracketed by HIP events (bench.py roofline leg)
enum { DX_PROF_CONV_GEMM = 0, DX_PROF_WGRAD_GEMM = 1, DX_PROF_ATTN_FWD = 2, DX_PROF_ATTN_BWD = 3,
       DX_PROF_UPSAMPLE = 4, DX_PROF_ROWS = 5, DX_PROF_NKINDS = 6 };

typedef float f32x4 __attribute__((ext_vector_type(4)));
// The 16-bit operand / storage type of the reduced-precision paths.  dx_gemm.hip, dx_ffpair.hip, dx_attention.hip and dx_rows.hip are
// compiled TWICE: as they stand (bf16: BASELINE.json config 2) and with -DDX_F16 (IEEE fp16: config 5), where every exported name
// gets the suffix _f16 (dx_f16_names.h) and every "bf16" flag / argument of the C ABI means "fp16".  Same kernels, same layouts,
// v_mfma_f32_16x16x32_f16 instead of _bf16.  (The vector typedefs keep their historic names.)
#ifdef DX_F16
typedef _Float16 dx_h16;
#define DX_MFMA_H16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16((A), (B), (C), 0, 0, 0)
#include "dx_f16_names.h"
#else
typedef __bf16 dx_h16;
#define DX_MFMA_H16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A), (B), (C), 0, 0, 0)
#endif
typedef dx_h16 bf16x8 __attribute__((ext_vector_type(8)));
typedef dx_h16 bf16x4 __attribute__((ext_vector_type(4)));

static inline int dx_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int dx_roundup(int a, int b) { return dx_cdiv(a, b) * b; }

#ifdef __HIPCC__
// ---- wave64 reductions -------------------------------------------------------------------------
__device__ __forceinline__ float dx_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float dx_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- counter-based dropout RNG -----------------------------------------------------------------
// Counter-based draw for (seed, index): 64 random bits = four 16-bit keep/drop decisions.  The same (seed, index) is
// evaluated again in the backward kernels, so no mask tensor is stored.  The attention kernels are VALU-bound with dropout on
// (a wave64 vector instruction takes 4 cycles; ~40 % of their instructions were this function), so the mixer is as short as the
// statistics allow: the high halves of index and seed are folded in by one multiply (loop-invariant in every caller: hoisted), the
// low word is a two-round multiply-xorshift finaliser (the "lowbias32" constants), the high word one more multiply-xorshift of the
// low one: 3 multiplies in the loop instead of 6, ~13 instructions instead of ~22.  Keep rate per field, cross-field, lag-1 along
// keys / rows / consecutive indices, seed vs seed + 1 correlations and byte uniformity of the keep mask are all at noise level on
// 4 M attention-shaped and 4 M consecutive indices (tests/test_host_glue_cpu.py restates the function in numpy and checks them).
__device__ __forceinline__ uint64_t dx_rand64(uint64_t seed, uint64_t idx) {
  uint32_t x = (uint32_t)idx ^ (uint32_t)seed;
  const uint32_t hi = (uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32);
  x ^= hi * 0x9E3779B1u;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  uint32_t y = (x ^ 0x85EBCA6Bu) * 0xC2B2AE35u; y ^= y >> 15;
  return ((uint64_t)y << 32) | x;
}
__device__ __forceinline__ float dx_dropout_scale(uint64_t seed, uint64_t elem, uint32_t thresh16, float inv_keep) {
  uint64_t r = dx_rand64(seed, elem >> 2);
  uint32_t bits = (uint32_t)(r >> (16 * (elem & 3))) & 0xFFFFu;
  return bits >= thresh16 ? inv_keep : 0.0f;
}
// Keep/drop selection without unpacking the draw: the 16-bit field is compared IN PLACE (SDWA word select) and the value is
// selected by the resulting mask -- one v_cmp + one v_cndmask per element instead of shift, mask, compare, select, multiply.
// The 1/(1-p) factor is not applied here: callers fold it into an operand or an epilogue scale (it is the same for all kept elements).
// result = (field >= thresh) ? x : alt, field = low (dx_keep_lo) or high (dx_keep_hi) half of w.  thresh < 65536.
// HAZARD RULE: x and alt must be results of ordinary VALU instructions, never MFMA accumulators -- the compiler's hazard recogniser
// does not look inside inline asm, and an asm that reads an accumulator too early reads stale data (callers subtract or copy first).
__device__ __forceinline__ float dx_keep_lo(uint32_t w, uint32_t thresh, float x, float alt) {
  float y;
  asm("v_cmp_ge_u16_sdwa vcc, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0\n\tv_cndmask_b32_e32 %0, %4, %3, vcc" : "=v"(y) : "v"(w), "v"(thresh), "v"(x), "v"(alt) : "vcc");
  return y;
}
__device__ __forceinline__ float dx_keep_hi(uint32_t w, uint32_t thresh, float x, float alt) {
  float y;
  asm("v_cmp_ge_u16_sdwa vcc, %1, %2 src0_sel:WORD_1 src1_sel:WORD_0\n\tv_cndmask_b32_e32 %0, %4, %3, vcc" : "=v"(y) : "v"(w), "v"(thresh), "v"(x), "v"(alt) : "vcc");
  return y;
}
// two values under the same decision (field already isolated in `bits`): x <- kept ? x : 0, y <- kept ? y : alt
__device__ __forceinline__ void dx_keep2(uint32_t bits, uint32_t thresh, float& x, float& y, float alt) {
  asm("v_cmp_ge_u32_e32 vcc, %2, %3\n\tv_cndmask_b32_e32 %0, 0, %0, vcc\n\tv_cndmask_b32_e32 %1, %4, %1, vcc" : "+v"(x), "+v"(y) : "v"(bits), "v"(thresh), "v"(alt) : "vcc");
}
// x[i] <- kept(i) ? x[i] : alt for the four consecutive elements of one draw
__device__ __forceinline__ void dx_keep4(uint64_t r, uint32_t thresh, float x[4], float alt) {
  const uint32_t lo = (uint32_t)r, hi = (uint32_t)(r >> 32);
  x[0] = dx_keep_lo(lo, thresh, x[0], alt); x[1] = dx_keep_hi(lo, thresh, x[1], alt);
  x[2] = dx_keep_lo(hi, thresh, x[2], alt); x[3] = dx_keep_hi(hi, thresh, x[3], alt);
}
// four consecutive elements (elem0 % 4 == 0)
__device__ __forceinline__ void dx_dropout_scale4(uint64_t seed, uint64_t elem0, uint32_t thresh16, float inv_keep, float out[4]) {
  uint64_t r = dx_rand64(seed, elem0 >> 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = ((uint32_t)(r >> (16 * i)) & 0xFFFFu) >= thresh16 ? inv_keep : 0.0f;
}
#endif

// Fused optimiser step over flat parameter / gradient buckets.
// Replaces torch.optim.Adam(betas, eps, weight_decay as L2-in-gradient, amsgrad=False) + clip_grad_norm_ as the reference
// trainer calls them (src/daft_exprt/train.py:278-280, :443-445): 14.4 M parameters x (read p, g, m, v; write p, m, v) =
// 0.4 GB of pure HBM traffic per step, one launch per bucket instead of ~10 ATen launches per parameter tensor.
#include "dx_common.h"
#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  __shared__ float part[4];
  float s = 0.f;
  const long n4 = n / 4;
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (; i + 3 * stride < n4; i += 4 * stride) {                 // four independent 16-byte loads in flight per thread
    const float4 v0 = reinterpret_cast<const float4*>(x)[i], v1 = reinterpret_cast<const float4*>(x)[i + stride];
    const float4 v2 = reinterpret_cast<const float4*>(x)[i + 2 * stride], v3 = reinterpret_cast<const float4*>(x)[i + 3 * stride];
    s += (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w);
    s1 += (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w);
    s2 += (v2.x * v2.x + v2.y * v2.y) + (v2.z * v2.z + v2.w * v2.w);
    s3 += (v3.x * v3.x + v3.y * v3.y) + (v3.z * v3.z + v3.w * v3.w);
  }
  for (; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  s = (s + s1) + (s2 + s3);
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += 256) s += x[i] * x[i];
  s = dx_wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

struct AdamArgs {
  float* p; const float* g; float* m; float* v; long n;
  float lr, b1, b2, eps, wd, bc1, bc2_sqrt;
  const float* normsq; float max_norm;
  float grad_scale;                 // gradients arrive multiplied by 1 / grad_scale (static loss scaling of the fp16 mode); 1 otherwise
  int* skipped;                     // optional device counter of skipped (non-finite gradient norm) updates
  float* norm_out;                  // optional: receives the (unscaled) global gradient norm, sqrt(*normsq) * grad_scale
  float* zero_after;                // optional: one float zeroed by this launch (the caller's OTHER squared-norm accumulator)
};

__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
  g = g * coef + a.wd * p;                             // clip, then L2 weight decay folded into the gradient (torch Adam)
  m = a.b1 * m + (1.f - a.b1) * g;
  v = a.b2 * v + (1.f - a.b2) * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= (a.lr / a.bc1) * (m / denom);
  return p;
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamArgs a) {
  float coef = a.grad_scale;
  if (a.normsq) {
    // Overflow guard (the fp16 mode's static loss scale can push a 16-bit gradient tensor to inf): a non-finite squared norm means at
    // least one non-finite gradient, and applying it would write NaN into every parameter and both moments for good.  The update is
    // skipped as a whole -- p, m, v untouched -- on every rank alike (the norm is taken after the all-reduce).
    const float nsq = *a.normsq;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (a.norm_out) *a.norm_out = sqrtf(nsq) * a.grad_scale;
      if (a.zero_after) *a.zero_after = 0.f;
    }
    if (!(nsq < INFINITY)) {
      if (a.skipped && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.skipped, 1);
      return;
    }
    if (a.max_norm < INFINITY) coef *= fminf(1.f, a.max_norm / (sqrtf(nsq) * a.grad_scale + 1e-6f));  // clip_grad_norm_
  }
  const long n4 = a.n / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 p = reinterpret_cast<float4*>(a.p)[i], m = reinterpret_cast<float4*>(a.m)[i], v = reinterpret_cast<float4*>(a.v)[i];
    const float4 g = reinterpret_cast<const float4*>(a.g)[i];
    adam_one(p.x, g.x, m.x, v.x, a, coef); adam_one(p.y, g.y, m.y, v.y, a, coef);
    adam_one(p.z, g.z, m.z, v.z, a, coef); adam_one(p.w, g.w, m.w, v.w, a, coef);
    reinterpret_cast<float4*>(a.p)[i] = p; reinterpret_cast<float4*>(a.m)[i] = m; reinterpret_cast<float4*>(a.v)[i] = v;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < a.n; i += 256) adam_one(a.p[i], a.g[i], a.m[i], a.v[i], a, coef);
}

}  // namespace

extern "C" {

// out[0] += sum x^2   (global gradient norm; out is caller-zeroed)
int dx_sumsq(const float* x, long n, float* out, void* stream) {
  DX_REQUIRE(x && out && n > 0 && ((uintptr_t)x % 16) == 0, "dx_sumsq: bad arguments");
  // every block ends in ONE atomic on the same address (~15 ns each, serialised): 2048 blocks made a 15 MB read take 31 us
  const int blocks = (int)std::min<long>((n / 4 + 255) / 256 + 1, 256);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, out);
  DX_LAUNCH_CHECK("dx_sumsq");
  return DX_OK;
}

// One Adam step on a flat bucket.  step >= 1.  normsq (optional, device scalar) = squared global gradient norm for clipping.
// grad_scale: every gradient (and the norm) is multiplied by it first (1 / loss scale when the backward ran on a scaled loss).
// skipped (optional, device int): when *normsq is not finite the update is skipped (p, m, v untouched) and *skipped is incremented.
// norm_out (optional): receives sqrt(*normsq) * grad_scale, what clip_grad_norm_ returns.  zero_after (optional): a float this launch
// sets to zero -- a caller that alternates between two squared-norm accumulators needs no fill launch per step (it must not be normsq).
int dx_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, const float* normsq, float max_norm, float grad_scale, int* skipped,
                 float* norm_out, float* zero_after, void* stream) {
  DX_REQUIRE(!zero_after || zero_after != normsq, "dx_adam_step: zero_after must not alias normsq");
  DX_REQUIRE(normsq || (!norm_out && !zero_after), "dx_adam_step: norm_out / zero_after need normsq");
  DX_REQUIRE(p && g && m && v && n > 0 && step >= 1, "dx_adam_step: bad arguments");
  DX_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
             "dx_adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  AdamArgs a{p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), normsq, max_norm, grad_scale, skipped, norm_out, zero_after};
  const int blocks = (int)std::min<long>((n / 4 + 255) / 256 + 1, 4096);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK("dx_adam_step");
  return DX_OK;
}

}  // extern "C"

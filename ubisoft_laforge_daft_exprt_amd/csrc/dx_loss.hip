// Fused loss reductions and their gradients (reference: src/daft_exprt/loss.py:99-146).
// mel L1 + L2 (length-normalised), energy consistency (L2 norm of exp(mel) over bins -> AvgPool1d(5,1,2) -> masked MSE),
// pitch consistency (masked MSE between the frozen predictor's output and the voiced ground truth).
// Predicted / target mels are (B, M, T) as in the reference; every kernel walks t fastest (coalesced), each tensor is read once
// per pass: HBM-bound, 2*B*M*T*4 bytes for the statistics pass and 3*B*M*T*4 for the gradient pass.
#include "dx_common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* scratch) {
  v = dx_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

// per (b, t) column: |d|, d^2 partial sums per utterance, energies of prediction and target.
// Block = 64 frames x 4 groups of mel bins (bins g, g+4, ...): a thread per frame walking all 80 bins gave 192 blocks of
// dependent loads for 256 CUs (43 us for 28 MB).
constexpr int MEL_TT = 64;
__global__ __launch_bounds__(256) void mel_stats_kernel(const float* __restrict__ mp, const float* __restrict__ mt,
                                                        float* __restrict__ ep, float* __restrict__ et,
                                                        float* __restrict__ l1sum, float* __restrict__ l2sum, int M, int T) {
  __shared__ float scratch[4];
  __shared__ float part[2][4][MEL_TT];
  const int b = blockIdx.y, tl = threadIdx.x & (MEL_TT - 1), mg = threadIdx.x / MEL_TT, t = blockIdx.x * MEL_TT + tl;
  float l1 = 0.f, l2 = 0.f, sp = 0.f, st = 0.f;
  if (t < T) {
#pragma unroll 4
    for (int m = mg; m < M; m += 4) {
      const size_t i = ((size_t)b * M + m) * T + t;
      const float p = mp[i], q = mt[i], d = p - q;
      l1 += fabsf(d); l2 += d * d;
      const float e1 = expf(p), e2 = expf(q);
      sp += e1 * e1; st += e2 * e2;
    }
  }
  part[0][mg][tl] = sp; part[1][mg][tl] = st;
  __syncthreads();
  if (mg == 0 && t < T) {
    ep[(size_t)b * T + t] = sqrtf((part[0][0][tl] + part[0][1][tl]) + (part[0][2][tl] + part[0][3][tl]));
    et[(size_t)b * T + t] = sqrtf((part[1][0][tl] + part[1][1][tl]) + (part[1][2][tl] + part[1][3][tl]));
  }
  const float s1 = block_sum_256(l1, scratch);
  const float s2 = block_sum_256(l2, scratch);
  if (threadIdx.x == 0) { atomicAdd(&l1sum[b], s1); atomicAdd(&l2sum[b], s2); }
}

// smoothed energy difference; des = 2 * (smooth(ep) - smooth(et)) * [t < len];  esum += diff^2 * mask
__global__ __launch_bounds__(256) void energy_diff_kernel(const float* __restrict__ ep, const float* __restrict__ et, const int* __restrict__ lens,
                                                          float* __restrict__ des, float* __restrict__ esum, int T) {
  __shared__ float scratch[4];
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  float e = 0.f;
  if (t < T) {
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int k = -2; k <= 2; ++k) {
      const int u = t + k;
      if (u >= 0 && u < T) { a += ep[(size_t)b * T + u]; c += et[(size_t)b * T + u]; }
    }
    const float diff = a / 5.f - c / 5.f;       // AvgPool1d counts the zero padding (count_include_pad=True)
    const bool valid = t < lens[b];
    des[(size_t)b * T + t] = valid ? 2.f * diff : 0.f;
    e = valid ? diff * diff : 0.f;
  }
  const float s = block_sum_256(e, scratch);
  if (threadIdx.x == 0 && s != 0.f) atomicAdd(esum, s);
}

// dmel[b,m,t] = c_l1/len_b * sign(d) + c_l2/len_b * 2 d + c_e * (sum_{|k|<=2} des[t+k] / 5) * exp(p)^2 / ep[t]
__global__ __launch_bounds__(256) void mel_grad_kernel(const float* __restrict__ mp, const float* __restrict__ mt,
                                                       const float* __restrict__ ep, const float* __restrict__ des, const int* __restrict__ lens,
                                                       float c_l1, float c_l2, float c_e, int e_per_total, float* __restrict__ dmel, int M, int T) {
  const int b = blockIdx.y, mg = threadIdx.x / MEL_TT, t = blockIdx.x * MEL_TT + (threadIdx.x & (MEL_TT - 1));
  if (e_per_total) {                                    // energy term normalised by the batch's total valid length (loss.py:129):
    __shared__ float scratch[4];                        // summed here, on the device, so that no host value is frozen into a graph
    float n = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.y; i += 256) n += (float)lens[i];
    c_e /= block_sum_256(n, scratch);
  }
  if (t >= T) return;
  const float inv_len = 1.f / (float)lens[b];
  float ge = 0.f;
  if (c_e != 0.f) {
    float a = 0.f;
#pragma unroll
    for (int k = -2; k <= 2; ++k) {
      const int u = t + k;
      if (u >= 0 && u < T) a += des[(size_t)b * T + u];
    }
    ge = c_e * (a / 5.f) / ep[(size_t)b * T + t];
  }
#pragma unroll 4
  for (int m = mg; m < M; m += 4) {                     // block = 64 frames x 4 groups of mel bins (see mel_stats_kernel)
    const size_t i = ((size_t)b * M + m) * T + t;
    const float p = mp[i], d = p - mt[i];
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    float g = (c_l1 * sgn + c_l2 * 2.f * d) * inv_len;
    if (c_e != 0.f) { const float e = expf(p); g += ge * e * e; }
    dmel[i] = g;
  }
}

// sums[0] += sum mask (pp - gt)^2, sums[1] += sum mask;  mask = t < len and gt != 0
__global__ __launch_bounds__(256) void pitch_mse_kernel(const float* __restrict__ pp, int ldp, const float* __restrict__ gt, const int* __restrict__ lens,
                                                        float* __restrict__ sums, int T) {
  __shared__ float scratch[4];
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  float e = 0.f, n = 0.f;
  if (t < T) {
    const float g = gt[(size_t)b * T + t];
    if (t < lens[b] && g != 0.f) { const float d = pp[((size_t)b * T + t) * ldp] - g; e = d * d; n = 1.f; }
  }
  const float se = block_sum_256(e, scratch);
  const float sn = block_sum_256(n, scratch);
  if (threadIdx.x == 0) { if (se != 0.f) atomicAdd(&sums[0], se); if (sn != 0.f) atomicAdd(&sums[1], sn); }
}

// dpp = scale * 2 (pp - gt) mask / (count + 1e-5)
__global__ __launch_bounds__(256) void pitch_grad_kernel(const float* __restrict__ pp, int ldp, const float* __restrict__ gt, const int* __restrict__ lens,
                                                         const float* __restrict__ sums, float scale, float* __restrict__ dpp, int ldd, int T) {
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float g = gt[(size_t)b * T + t];
  float v = 0.f;
  if (t < lens[b] && g != 0.f) v = scale * 2.f * (pp[((size_t)b * T + t) * ldp] - g) / (sums[1] + 1e-5f);
  dpp[((size_t)b * T + t) * ldd] = v;
}

// The seven loss terms and the total (loss.py:85-157) from the reductions above, plus the two small gradients, in one block:
// replaces ~30 one-element ATen launches per step.
struct LossFinalizeArgs {
  const float* ce; const float* spk_w_dev; float spk_w;
  const float* dlogits; float* d_spk; int n_logits;
  const float* pm; float* d_pm; int n_pm; float pmw;
  const float* l1sum; const float* l2sum; const int* lens; int B; int M; float msw;
  const float* esum; float ecw; const float* psum; float pcw;
  float* terms; float* total;
  float grad_scale;   // the two small gradients leave multiplied by it (a trainer's loss scale / accumulation factor); terms and total do not
};
__global__ __launch_bounds__(256) void loss_finalize_kernel(const LossFinalizeArgs a) {
  __shared__ float scratch[4];
  const int tid = threadIdx.x;
  const float w = a.spk_w_dev ? *a.spk_w_dev : a.spk_w;
  for (int i = tid; i < a.n_logits; i += 256) a.d_spk[i] = a.dlogits[i] * (w * a.grad_scale);
  float sq = 0.f;
  for (int i = tid; i < a.n_pm; i += 256) sq += a.pm[i] * a.pm[i];
  const float nrm = sqrtf(block_sum_256(sq, scratch));
  for (int i = tid; i < a.n_pm; i += 256) a.d_pm[i] = a.grad_scale * a.pmw * a.pm[i] / nrm;
  float l1 = 0.f, l2 = 0.f, n = 0.f;
  for (int i = tid; i < a.B; i += 256) {
    const float len = (float)a.lens[i], d = (float)a.M * len;
    l1 += a.l1sum[i] / d; l2 += a.l2sum[i] / d; n += len;
  }
  l1 = block_sum_256(l1, scratch);
  l2 = block_sum_256(l2, scratch);
  n = block_sum_256(n, scratch);
  if (tid == 0) {
    const float ce = a.ce ? *a.ce : 0.f;
    const float t0 = w * ce, t2 = a.n_pm ? a.pmw * nrm : 0.f, t3 = a.msw * l1 / (float)a.B, t4 = a.msw * l2 / (float)a.B;
    const float t5 = a.esum ? *a.esum / n : 0.f, t6 = a.psum ? a.psum[0] / (a.psum[1] + 1e-5f) : 0.f;
    a.terms[0] = t0; a.terms[1] = ce; a.terms[2] = t2; a.terms[3] = t3; a.terms[4] = t4; a.terms[5] = t5; a.terms[6] = t6;
    *a.total = t0 + t2 + t3 + t4 + a.ecw * t5 + a.pcw * t6;
  }
}

}  // namespace

extern "C" {

// l1sum, l2sum [B] and ep, et [B][T]; l1sum/l2sum caller-zeroed
int dx_mel_stats(const float* mel_pred, const float* mel_target, float* ep, float* et, float* l1sum, float* l2sum,
                 int B, int M, int T, void* stream) {
  DX_REQUIRE(mel_pred && mel_target && ep && et && l1sum && l2sum && B > 0 && M > 0 && T > 0, "dx_mel_stats: bad arguments");
  hipLaunchKernelGGL(mel_stats_kernel, dim3(dx_cdiv(T, MEL_TT), B), dim3(256), 0, (hipStream_t)stream, mel_pred, mel_target, ep, et, l1sum, l2sum, M, T);
  DX_LAUNCH_CHECK("dx_mel_stats");
  return DX_OK;
}

int dx_energy_diff(const float* ep, const float* et, const int* lens, float* des, float* esum, int B, int T, void* stream) {
  DX_REQUIRE(ep && et && lens && des && esum && B > 0 && T > 0, "dx_energy_diff: bad arguments");
  hipLaunchKernelGGL(energy_diff_kernel, dim3(dx_cdiv(T, 256), B), dim3(256), 0, (hipStream_t)stream, ep, et, lens, des, esum, T);
  DX_LAUNCH_CHECK("dx_energy_diff");
  return DX_OK;
}

int dx_mel_grad(const float* mel_pred, const float* mel_target, const float* ep, const float* des, const int* lens,
                float c_l1, float c_l2, float c_e, int e_per_total, float* dmel, int B, int M, int T, void* stream) {
  DX_REQUIRE(mel_pred && mel_target && lens && dmel && B > 0 && M > 0 && T > 0, "dx_mel_grad: bad arguments");
  DX_REQUIRE(c_e == 0.f || (ep && des), "dx_mel_grad: energy term needs ep and des");
  hipLaunchKernelGGL(mel_grad_kernel, dim3(dx_cdiv(T, MEL_TT), B), dim3(256), 0, (hipStream_t)stream, mel_pred, mel_target, ep, des, lens, c_l1, c_l2, c_e, e_per_total, dmel, M, T);
  DX_LAUNCH_CHECK("dx_mel_grad");
  return DX_OK;
}

// terms[7] = {speaker_loss, speaker_ce_raw, post_mult_loss, mel_l1, mel_l2, energy_consistency, pitch_consistency}, total[1] =
// their weighted sum; d_spk = dlogits * w, d_pm = pmw * pm / ||pm||.  w = *spk_w_dev when given (a device scalar a captured graph
// re-reads on every replay), else spk_w.  NULL ce / pm / esum / psum switch the corresponding term off.
int dx_loss_finalize(const float* ce, const float* spk_w_dev, float spk_w, const float* dlogits, float* d_spk, int n_logits,
                     const float* pm, float* d_pm, int n_pm, float pmw,
                     const float* l1sum, const float* l2sum, const int* lens, int B, int M, float msw,
                     const float* esum, float ecw, const float* psum, float pcw, float* terms, float* total, float grad_scale, void* stream) {
  DX_REQUIRE(l1sum && l2sum && lens && terms && total && B > 0 && M > 0, "dx_loss_finalize: bad arguments");
  DX_REQUIRE(n_logits == 0 || (dlogits && d_spk && ce), "dx_loss_finalize: speaker term needs ce, dlogits, d_spk");
  DX_REQUIRE(n_pm == 0 || (pm && d_pm), "dx_loss_finalize: post-multiplier term needs pm, d_pm");
  LossFinalizeArgs a{ce, spk_w_dev, spk_w, dlogits, d_spk, n_logits, pm, d_pm, n_pm, pmw, l1sum, l2sum, lens, B, M, msw, esum, ecw, psum, pcw, terms, total, grad_scale};
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK("dx_loss_finalize");
  return DX_OK;
}

int dx_pitch_mse(const float* pp, int ldp, const float* gt, const int* lens, float* sums, int B, int T, void* stream) {
  DX_REQUIRE(pp && ldp >= 1 && gt && lens && sums && B > 0 && T > 0, "dx_pitch_mse: bad arguments");
  hipLaunchKernelGGL(pitch_mse_kernel, dim3(dx_cdiv(T, 256), B), dim3(256), 0, (hipStream_t)stream, pp, ldp, gt, lens, sums, T);
  DX_LAUNCH_CHECK("dx_pitch_mse");
  return DX_OK;
}

int dx_pitch_grad(const float* pp, int ldp, const float* gt, const int* lens, const float* sums, float scale, float* dpp, int ldd, int B, int T, void* stream) {
  DX_REQUIRE(pp && ldp >= 1 && gt && lens && sums && dpp && ldd >= 1 && B > 0 && T > 0, "dx_pitch_grad: bad arguments");
  hipLaunchKernelGGL(pitch_grad_kernel, dim3(dx_cdiv(T, 256), B), dim3(256), 0, (hipStream_t)stream, pp, ldp, gt, lens, sums, scale, dpp, ldd, T);
  DX_LAUNCH_CHECK("dx_pitch_grad");
  return DX_OK;
}

}  // extern "C"

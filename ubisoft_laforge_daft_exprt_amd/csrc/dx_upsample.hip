// Gaussian upsampling (reference: src/daft_exprt/model.py:417-510) without the B*L*D*T broadcast.
//
//   prep     xs[b,l,:] = enc + conv1->D(energy) + conv1->D(pitch);  z = (xs + conv1->D(dur)) . w_r + b_r
//            sigma = max(l < len ? softplus(z) : 1, 1e-3);  mu = dur_int/2 + exclusive_cumsum(dur_int)   (int64 -> exact)
//   weights  p[l,t] = exp(-((t+.5-mu)^2)/(2 sigma^2) - log sigma - log sqrt(2 pi)), 0 for l >= len
//            w = p / (sum_l p + 1e-20)  -> weights (B, L, T) (returned tensor), x_up[b,t,:] = sum_l w[l,t] xs[b,l,:]
//   backward dxs, dsigma via w (d log p), then the symbol-level parameter gradients.
//
// HBM traffic (fp32): reads B*L*D + 4*B*L, writes B*T*D + B*L*T  (35 MB at B=48, L=120, T=840; the reference
// materialises 2 GB).  One workgroup owns TT frames of one utterance and keeps the [L][TT] weight tile in LDS.
#include "dx_common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int D = 128;
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;

__device__ __forceinline__ float softplus_ref(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// one wave per utterance: exclusive cumsum of integer durations (int64, exact), means, total frames.  (A single thread walking the row
// paid one dependent-load latency per symbol: 17 us for L = 120.)
__global__ __launch_bounds__(64) void dur_scan_kernel(const long* __restrict__ dur_int, float* __restrict__ mu, long* __restrict__ totals, int L) {
  const int b = blockIdx.x, lane = threadIdx.x;
  long carry = 0;
  for (int base = 0; base < L; base += 64) {
    const int l = base + lane;
    const long d = l < L ? dur_int[(size_t)b * L + l] : 0;
    long inc = d;                                            // inclusive prefix over the 64 lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const long v = __shfl_up(inc, off, 64);
      if (lane >= off) inc += v;
    }
    if (l < L) mu[(size_t)b * L + l] = (float)d / 2.f + (float)(carry + inc - d);   // dur_int.float()/2 + cumsum[:-1] (model.py:485-487)
    carry += __shfl(inc, 63, 64);
  }
  if (lane == 0) totals[b] = carry;
}

struct PrepArgs {
  const float* enc; const float* dur; const float* energy; const float* pitch;
  const float* wd; const float* bd; const float* we; const float* be; const float* wp; const float* bp;
  const float* wr; const float* br;
  const int* lens;
  float* xs; float* z; float* sigma;
  int B, L;
};

// wave per symbol row; lane owns channels 2*lane, 2*lane+1
__global__ __launch_bounds__(256) void upsample_prep_kernel(const PrepArgs a) {
  const int lane = threadIdx.x & 63, c = lane * 2;
  float wD[2][3], wE[2][3], wP[2][3], bD[2], bE[2], bP[2], wR[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int t = 0; t < 3; ++t) { wD[k][t] = a.wd[(c + k) * 3 + t]; wE[k][t] = a.we[(c + k) * 3 + t]; wP[k][t] = a.wp[(c + k) * 3 + t]; }
    bD[k] = a.bd[c + k]; bE[k] = a.be[c + k]; bP[k] = a.bp[c + k]; wR[k] = a.wr[c + k];
  }
  const float br = a.br[0];
  const long rows = (long)a.B * a.L;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / a.L), l = (int)(row - (long)b * a.L);
    const float* db = a.dur + (size_t)b * a.L; const float* eb = a.energy + (size_t)b * a.L; const float* pb = a.pitch + (size_t)b * a.L;
    const bool lo = l > 0, hi = l + 1 < a.L;
    const float d0 = lo ? db[l - 1] : 0.f, d1 = db[l], d2 = hi ? db[l + 1] : 0.f;
    const float e0 = lo ? eb[l - 1] : 0.f, e1 = eb[l], e2 = hi ? eb[l + 1] : 0.f;
    const float p0 = lo ? pb[l - 1] : 0.f, p1 = pb[l], p2 = hi ? pb[l + 1] : 0.f;
    const float2 x = *reinterpret_cast<const float2*>(a.enc + row * D + c);
    float xv[2] = {x.x, x.y}, dot = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float ev = bE[k] + (wE[k][0] * e0 + wE[k][1] * e1 + wE[k][2] * e2);
      const float pv = bP[k] + (wP[k][0] * p0 + wP[k][1] * p1 + wP[k][2] * p2);
      const float dv = bD[k] + (wD[k][0] * d0 + wD[k][1] * d1 + wD[k][2] * d2);
      xv[k] = (xv[k] + ev) + pv;                       // model.py:470
      dot += (xv[k] + dv) * wR[k];                     // model.py:475-476
    }
    *reinterpret_cast<float2*>(a.xs + row * D + c) = make_float2(xv[0], xv[1]);
    dot = dx_wave_sum(dot) + br;
    if (lane == 0) {
      a.z[row] = dot;
      const float rng = l < a.lens[b] ? softplus_ref(dot) : 1.f;
      a.sigma[row] = fmaxf(rng, 1e-3f);
    }
  }
}

struct UpArgs {
  const float* xs; const float* mu; const float* sigma; const int* lens;
  float* weights;      // [B][L][T]
  float* xup;          // [B][T][D]
  int B, L, T;
};

// grid (ceil(T/TT), B), 256 threads.  LDS: wt[L][TT] + colsum[4][TT]
template <int TT>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const UpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wt = smem;                       // [L][TT]
  float* part = smem + (size_t)a.L * TT;  // [256 / TT][TT]
  float* lgs = part + 256;                // [L] log(sigma_l)
  float* two = lgs + a.L;                 // [L] 2 sigma_l^2
  const int b = blockIdx.y, t0 = blockIdx.x * TT;
  const int tid = threadIdx.x;
  const int len = a.lens[b];
  const float* mu = a.mu + (size_t)b * a.L;
  const float* sg = a.sigma + (size_t)b * a.L;
  // per-symbol terms of the log-density once per workgroup instead of once per (symbol, frame): the same values, so the same bits
  for (int l = tid; l < a.L; l += 256) { const float sd = sg[min(l, max(len, 1) - 1)]; lgs[l] = logf(sd); two[l] = 2.f * (sd * sd); }
  __syncthreads();
  // pass 1: probabilities into LDS, column sums
  {
    constexpr int LG = 256 / TT;          // symbol groups walking l in parallel
    const int tt = tid % TT, lg = tid / TT;
    const float tv = (float)(t0 + tt) + 0.5f;
    float s = 0.f;
    for (int l = lg; l < a.L; l += LG) {
      float p = 0.f;
      if (l < len) {
        const float d = tv - mu[l];
        p = expf(-(d * d) / two[l] - lgs[l] - LOG_SQRT_2PI);
      }
      wt[(size_t)l * TT + tt] = p;
      s += p;
    }
    // reduce the LG partial sums per frame through LDS
    __syncthreads();
    float* red = part;                    // reuse as [LG][TT] (LG <= 16 -> need LG*TT floats)
    red[lg * TT + tt] = s;
    __syncthreads();
    if (lg == 0) {
      float tot = 0.f;
      for (int k = 0; k < LG; ++k) tot += red[k * TT + tt];
      red[tt] = tot + 1e-20f;
    }
    __syncthreads();
    const float den = red[tt];
    for (int l = lg; l < a.L; l += LG) {
      const float w = wt[(size_t)l * TT + tt] / den;   // model.py:505
      wt[(size_t)l * TT + tt] = w;
      if (t0 + tt < a.T) a.weights[((size_t)b * a.L + l) * a.T + t0 + tt] = w;
    }
    __syncthreads();
  }
  // pass 2: x_up[t][c] = sum_l w[l][t] * xs[l][c]
  if constexpr (TT == 16 || TT == 32) {
    // on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, exact fp32 products): A = xs^T (m = channel, k = symbol), B = w (k = symbol,
    // n = frame), both operands read as they lie (row-major xs from L1 / L2, the probability tile from LDS) - one scalar each per MFMA.
    // A wave owns 32 channels x the tile's frames; two accumulator chains per 16 x 16 tile.  (As FMA loops: 120 iterations of one
    // global load, two LDS reads and eight FMAs per thread.)
    constexpr int NT = TT / 16;
    const int lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const float* xb = a.xs + (size_t)b * a.L * D;
    const int lmax = min(len, a.L);
    const int nk = ((lmax + 3) >> 2) + 1 & ~1;                     // k steps of 4 symbols, rounded up to a pair
    f32x4 acc[2][NT][2];                                           // [channel tile][frame tile][chain]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) { acc[i][j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][j][1] = acc[i][j][0]; }
    const int c0 = wave * 32 + r;
    for (int kb = 0; kb < nk; kb += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int l = (kb + h) * 4 + g;
        const bool in = l < lmax;
        const int q = min(l, max(lmax, 1) - 1);                    // clamped: no load sits in a branch
        const float x0 = xb[(size_t)q * D + c0], x1 = xb[(size_t)q * D + c0 + 16];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float wv = wt[(size_t)q * TT + j * 16 + r];
          const float bw = in ? wv : 0.f;
          acc[0][j][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, bw, acc[0][j][h], 0, 0, 0);
          acc[1][j][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, bw, acc[1][j][h], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = t0 + j * 16 + r;                                // D[m = channel 4g + e][n = frame r]
      if (t < a.T) {
        float* orow = a.xup + ((size_t)b * a.T + t) * D + wave * 32 + g * 4;
        *reinterpret_cast<f32x4*>(orow) = acc[0][j][0] + acc[0][j][1];
        *reinterpret_cast<f32x4*>(orow + 16) = acc[1][j][0] + acc[1][j][1];
      }
    }
  } else {
    constexpr int TH = TT / 2;
    const int c = tid & 127, th = tid >> 7;
    float acc[TH];
#pragma unroll
    for (int k = 0; k < TH; ++k) acc[k] = 0.f;
    const float* xb = a.xs + (size_t)b * a.L * D;
    for (int l = 0; l < len; ++l) {
      const float xv = xb[(size_t)l * D + c];
      const float* wrow = wt + (size_t)l * TT + th * TH;
#pragma unroll
      for (int k = 0; k < TH; k += 4) {
        const float4 w4 = *reinterpret_cast<const float4*>(wrow + k);
        acc[k] += w4.x * xv; acc[k + 1] += w4.y * xv; acc[k + 2] += w4.z * xv; acc[k + 3] += w4.w * xv;
      }
    }
#pragma unroll
    for (int k = 0; k < TH; ++k) {
      const int t = t0 + th * TH + k;
      if (t < a.T) a.xup[((size_t)b * a.T + t) * D + c] = acc[k];
    }
  }
}

struct UpBwdArgs {
  const float* dxup;   // [B][T][D]
  const float* xs; const float* mu; const float* sigma; const float* weights; const int* lens;
  float* dxs;          // [B][L][D] accumulated (atomics, caller-zeroed)
  float* dsigma;       // [B][L]    accumulated (atomics, caller-zeroed)
  int B, L, T;
};

// four v_mfma_f32_16x16x4_f32 over 16 consecutive k: a / b = the lane's 4 consecutive k values (k = 4 (lane / 16) + e) of row (lane % 16)
__device__ __forceinline__ f32x4 up_mma4(const float4& a, const float4& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}

// grid (ceil(T/TT), B), 256 threads.  LDS: g[TT][D+4] (dx_up tile), dw[Lp][TT], w[Lp][TT+4], inner[256], and one region that holds the
// symbol rows xs[32][D+4] during the first product and the transposed gradient tile gT[D][TT+4] during the second (Lp = L rounded up to 16).
// Both products run on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 multiply-add, so the f32 parity mode keeps its bar):
//   dw[l][t] = sum_c xs[l][c] g[t][c]   (A = xs rows, B = g rows, k = c)        dxs[l][c] += sum_t w[l][t] g[t][c]   (A = w rows, B = gT rows, k = t)
// As fp32 FMA loops over LDS operands they were 2.6 GFLOP of VALU work per launch behind two LDS reads per 4 FMAs: 99 us.
// (Tried: one workgroup walking 4 consecutive frame tiles with the d(xs) tiles kept in registers, a quarter of the atomics: 71 -> 137 us.
// The kernel is bound by the serial phases of ONE workgroup per tile, not by its atomics; fewer, longer workgroups made that worse.)
template <int TT>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const UpBwdArgs a) {
  constexpr int GLD = D + 4;
  constexpr int WLD = TT + 4;
  constexpr int DLD = TT + 1;                          // dw rows: odd stride (thread-per-symbol reads down a column)
  constexpr int XB = 32;
  constexpr int NTT = TT / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Lp = (a.L + 15) & ~15;
  float* gt = smem;                                   // [TT][GLD]
  float* dwt = gt + TT * GLD;                         // [Lp][DLD]
  float* wt = dwt + (((size_t)Lp * DLD + 3) & ~(size_t)3);   // [Lp][WLD]
  float* inner = wt + (size_t)Lp * WLD;               // [256] scratch
  float* xst = inner + 256;                           // [XB][GLD] symbol rows of the current block, later gT [D][WLD]
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g4 = (lane >> 4) * 4;
  const int len = a.lens[b];
  const float* xb = a.xs + (size_t)b * a.L * D;
  const int nlt = (min(len, a.L) + 15) >> 4;
  const int t0 = blockIdx.x * TT;
  // stage dx_up tile (zero beyond T) and the weight tile
  bool nonzero = false;
  for (int u = tid; u < TT * (D / 4); u += 256) {
    const int tt = u / (D / 4), q = u % (D / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t0 + tt < a.T) v = *reinterpret_cast<const float4*>(a.dxup + ((size_t)b * a.T + t0 + tt) * D + q * 4);
    nonzero |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    *reinterpret_cast<float4*>(gt + tt * GLD + q * 4) = v;
  }
  // frame tiles beyond the utterance carry an all-zero gradient (the decoder masks them): every term below is then zero
  if (!__syncthreads_or(nonzero)) return;
  for (int u = tid; u < Lp * TT; u += 256) {
    const int l = u / TT, tt = u % TT;
    wt[l * WLD + tt] = (l < len && t0 + tt < a.T) ? a.weights[((size_t)b * a.L + l) * a.T + t0 + tt] : 0.f;
    dwt[l * DLD + tt] = 0.f;                                     // symbol rows the first product does not reach (l >= len rounded up to 32)
  }
  // ---- dw[l][t]: the symbol rows go through LDS 32 at a time; a 32-row block is 2 x NTT output tiles of 16 x 16, handed to the waves in turn
  for (int l0 = 0; l0 < len; l0 += XB) {
    __syncthreads();
    for (int u = tid; u < XB * (D / 4); u += 256) {
      const int l = u / (D / 4), q = u % (D / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (l0 + l < len) v = *reinterpret_cast<const float4*>(xb + (size_t)(l0 + l) * D + q * 4);
      *reinterpret_cast<float4*>(xst + l * GLD + q * 4) = v;
    }
    __syncthreads();
    for (int tile = wave; tile < 2 * NTT; tile += 4) {
      const int lt = tile / NTT, tt2 = tile - lt * NTT;
      if (l0 + lt * 16 >= Lp) continue;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc;   // two chains: a 16x16x4 MFMA waits for the previous one's result on the same accumulator
      const float* ap = xst + (lt * 16 + r) * GLD + g4;
      const float* bp = gt + (tt2 * 16 + r) * GLD + g4;
#pragma unroll
      for (int kb = 0; kb < D / 16; kb += 2) {
        acc = up_mma4(*reinterpret_cast<const float4*>(ap + kb * 16), *reinterpret_cast<const float4*>(bp + kb * 16), acc);
        acc1 = up_mma4(*reinterpret_cast<const float4*>(ap + kb * 16 + 16), *reinterpret_cast<const float4*>(bp + kb * 16 + 16), acc1);
      }
      acc += acc1;
#pragma unroll
      for (int e = 0; e < 4; ++e) dwt[(size_t)(l0 + lt * 16 + g4 + e) * DLD + tt2 * 16 + r] = acc[e];     // D[m = 4 (lane / 16) + e][n = lane % 16]
    }
  }
  __syncthreads();
  // gT[c][t] for the second product (the symbol-row region is free now), and inner[t] = sum_l dw[l][t] w[l][t]
  float* gT = xst;
  for (int u = tid; u < TT * D; u += 256) {
    const int tt = u / D, c = u - tt * D;
    gT[c * WLD + tt] = gt[tt * GLD + c];
  }
  constexpr int LG = 256 / TT;
  {
    const int tt = tid % TT, lg = tid / TT;
    float inn = 0.f;
    for (int l = lg; l < len; l += LG) inn += dwt[(size_t)l * DLD + tt] * wt[l * WLD + tt];
    inner[lg * TT + tt] = inn;
  }
  __syncthreads();
  if (tid < TT) {
    float tot = 0.f;
    for (int k = 0; k < LG; ++k) tot += inner[k * TT + tid];
    inner[tid] = tot;
  }
  __syncthreads();
  // d log p[l][t] = w (dw - inner[t]);  dsigma[l] += sum_t dlogp ((t+.5-mu)^2 / sigma^3 - 1/sigma)
  // thread per symbol, frames in a loop (a wave per symbol with lanes over the 32 frames left half the lanes idle and paid a 6-step wave
  // reduction per symbol: ~9 k cycles of this workgroup's ~50 k)
  for (int l = tid; l < len; l += 256) {
    const float m = a.mu[(size_t)b * a.L + l], sd = a.sigma[(size_t)b * a.L + l];
    const float i3 = 1.f / (sd * sd * sd), i1 = 1.f / sd;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < TT; ++k) {
      const float dlp = wt[l * WLD + k] * (dwt[(size_t)l * DLD + k] - inner[k]);
      const float d = ((float)(t0 + k) + 0.5f) - m;
      s += dlp * ((d * d) * i3 - i1);
    }
    if (s != 0.f) atomicAdd(&a.dsigma[(size_t)b * a.L + l], s);
  }
  // ---- dxs[l][c] += sum_t w[l][t] g[t][c]: (Lp / 16) x 8 output tiles, k = TT
  {
    for (int tile = wave; tile < nlt * (D / 16); tile += 4) {
      const int lt = tile / (D / 16), ct = tile - lt * (D / 16);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* ap = wt + (lt * 16 + r) * WLD + g4;
      const float* bp = gT + (ct * 16 + r) * WLD + g4;
#pragma unroll
      for (int kb = 0; kb < TT / 16; ++kb)
        acc = up_mma4(*reinterpret_cast<const float4*>(ap + kb * 16), *reinterpret_cast<const float4*>(bp + kb * 16), acc);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int l = lt * 16 + g4 + e;
        if (l < len && acc[e] != 0.f) atomicAdd(&a.dxs[((size_t)b * a.L + l) * D + ct * 16 + r], acc[e]);
      }
    }
  }
}

struct SymBwdArgs {
  const float* dxs_in;   // [B][L][D] gradient w.r.t. xs from the frame pass
  const float* dsigma;   // [B][L]
  const float* xs; const float* z; const float* dur; const int* lens;
  const float* wd; const float* bd; const float* wr;
  float* dxs_out;        // [B][L][D] total gradient w.r.t. xs (== d enc == d energy-proj == d pitch-proj outputs)
  float* dz;             // [B][L]
  float* dwr; float* dbr;  // [D], [1] accumulated
  int B, L;
};

// wave per symbol row
__global__ __launch_bounds__(256) void upsample_sym_bwd_kernel(const SymBwdArgs a) {
  __shared__ float red[4][D];
  __shared__ float redb[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane * 2;
  float wD[2][3], bD[2], wR[2], gwr[2] = {0.f, 0.f}, gbr = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int t = 0; t < 3; ++t) wD[k][t] = a.wd[(c + k) * 3 + t];
    bD[k] = a.bd[c + k]; wR[k] = a.wr[c + k];
  }
  const long rows = (long)a.B * a.L;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / a.L), l = (int)(row - (long)b * a.L);
    float dzv = 0.f;
    if (l < a.lens[b]) {
      const float zv = a.z[row];
      const float rng = softplus_ref(zv);
      // clamp(min=1e-3) passes the gradient where the input is >= min; softplus' = sigmoid (1 beyond the threshold 20)
      if (rng >= 1e-3f) dzv = a.dsigma[row] * (zv > 20.f ? 1.f : 1.f / (1.f + expf(-zv)));
    }
    const float* db = a.dur + (size_t)b * a.L;
    const float d0 = l > 0 ? db[l - 1] : 0.f, d1 = db[l], d2 = l + 1 < a.L ? db[l + 1] : 0.f;
    const float2 g = *reinterpret_cast<const float2*>(a.dxs_in + row * D + c);
    const float2 x = *reinterpret_cast<const float2*>(a.xs + row * D + c);
    const float gv[2] = {g.x, g.y}, xv[2] = {x.x, x.y};
    float o[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float dv = bD[k] + (wD[k][0] * d0 + wD[k][1] * d1 + wD[k][2] * d2);
      gwr[k] += dzv * (xv[k] + dv);
      o[k] = gv[k] + dzv * wR[k];
    }
    gbr += dzv;
    *reinterpret_cast<float2*>(a.dxs_out + row * D + c) = make_float2(o[0], o[1]);
    if (lane == 0) a.dz[row] = dzv;
  }
  red[wave][c] = gwr[0]; red[wave][c + 1] = gwr[1];
  if (lane == 0) redb[wave] = gbr;
  __syncthreads();
  if (threadIdx.x < D) atomicAdd(&a.dwr[threadIdx.x], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
  if (threadIdx.x == 0) atomicAdd(&a.dbr[0], (redb[0] + redb[1]) + (redb[2] + redb[3]));
}

template <int TT> size_t fwd_smem(int L) { return ((size_t)L * TT + (size_t)(256 / TT) * TT + 2 * (size_t)L) * sizeof(float); }
template <int TT> size_t bwd_smem(int L) {
  const size_t Lp = (size_t)((L + 15) & ~15);
  return ((size_t)TT * (D + 4) + ((Lp * (TT + 1) + 3) & ~(size_t)3) + Lp * (TT + 4) + 256 + std::max<size_t>(32 * (D + 4), (size_t)D * (TT + 4))) * sizeof(float);
}

template <int TT>
int launch_fwd(const UpArgs& a, hipStream_t s) {
  const size_t smem = fwd_smem<TT>(a.L);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&upsample_fwd_kernel<TT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(upsample_fwd_kernel<TT>, dim3(dx_cdiv(a.T, TT), a.B), dim3(256), smem, s, a);
  return DX_OK;
}
template <int TT>
int launch_bwd(const UpBwdArgs& a, hipStream_t s) {
  const size_t smem = bwd_smem<TT>(a.L);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&upsample_bwd_kernel<TT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(upsample_bwd_kernel<TT>, dim3(dx_cdiv(a.T, TT), a.B), dim3(256), smem, s, a);
  return DX_OK;
}

constexpr size_t LDS_BUDGET = 150 * 1024;

}  // namespace

extern "C" {

// mu[b,l] (fp32, exact for totals < 2^24) and totals[b] = sum_l dur_int[b,l] (int64)
int dx_duration_scan(const long* dur_int, float* mu, long* totals, int B, int L, void* stream) {
  DX_REQUIRE(dur_int && mu && totals && B > 0 && L > 0, "dx_duration_scan: bad arguments");
  hipLaunchKernelGGL(dur_scan_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, dur_int, mu, totals, L);
  DX_LAUNCH_CHECK("dx_duration_scan");
  return DX_OK;
}

int dx_upsample_prep(const float* enc, const float* dur, const float* energy, const float* pitch,
                     const float* wd, const float* bd, const float* we, const float* be, const float* wp, const float* bp,
                     const float* wr, const float* br, const int* lens, float* xs, float* z, float* sigma,
                     int B, int L, int Dm, void* stream) {
  DX_REQUIRE(enc && dur && energy && pitch && wd && bd && we && be && wp && bp && wr && br && lens && xs && z && sigma, "dx_upsample_prep: null pointer");
  DX_REQUIRE(Dm == D && B > 0 && L > 0, "dx_upsample_prep: hidden dim must be 128 (got %d)", Dm);
  PrepArgs a{enc, dur, energy, pitch, wd, bd, we, be, wp, bp, wr, br, lens, xs, z, sigma, B, L};
  const long rows = (long)B * L;
  hipLaunchKernelGGL(upsample_prep_kernel, dim3((int)std::min<long>((rows + 3) / 4, 4096)), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK("dx_upsample_prep");
  return DX_OK;
}

int dx_upsample_fwd(const float* xs, const float* mu, const float* sigma, const int* lens, float* weights, float* xup,
                    int B, int L, int T, int Dm, void* stream) {
  DX_REQUIRE(xs && mu && sigma && lens && weights && xup, "dx_upsample_fwd: null pointer");
  DX_REQUIRE(Dm == D && B > 0 && L > 0 && T > 0, "dx_upsample_fwd: bad dims (D must be 128)");
  UpArgs a{xs, mu, sigma, lens, weights, xup, B, L, T};
  hipStream_t s = (hipStream_t)stream;
  dx_prof_begin(DX_PROF_UPSAMPLE, s);
  static const int force_tt = getenv("DX_UP_FWD_TT") ? atoi(getenv("DX_UP_FWD_TT")) : 16;   // 64 / 32 / 16-frame tiles: 39.5 / 30.9 / 26.6 us
  if (force_tt == 64 && fwd_smem<64>(L) <= LDS_BUDGET) launch_fwd<64>(a, s);
  else if (force_tt != 16 && fwd_smem<32>(L) <= LDS_BUDGET) launch_fwd<32>(a, s);
  else { DX_REQUIRE(fwd_smem<16>(L) <= LDS_BUDGET, "dx_upsample_fwd: L=%d too long for the LDS weight tile", L); launch_fwd<16>(a, s); }
  dx_prof_end(DX_PROF_UPSAMPLE, s);
  DX_LAUNCH_CHECK("dx_upsample_fwd");
  return DX_OK;
}

int dx_upsample_bwd(const float* dxup, const float* xs, const float* mu, const float* sigma, const float* weights, const int* lens,
                    float* dxs, float* dsigma, int B, int L, int T, int Dm, void* stream) {
  DX_REQUIRE(dxup && xs && mu && sigma && weights && lens && dxs && dsigma, "dx_upsample_bwd: null pointer");
  DX_REQUIRE(Dm == D && B > 0 && L > 0 && T > 0, "dx_upsample_bwd: bad dims (D must be 128)");
  UpBwdArgs a{dxup, xs, mu, sigma, weights, lens, dxs, dsigma, B, L, T};
  hipStream_t s = (hipStream_t)stream;
  // 32-frame tiles: 65 KB of LDS at L = 120, two workgroups per CU (64-frame tiles: 112 KB, one per CU, 153 vs 99 us)
  static const int force_tt = getenv("DX_UP_TT") ? atoi(getenv("DX_UP_TT")) : 32;
  if (force_tt == 64 && bwd_smem<64>(L) <= LDS_BUDGET) launch_bwd<64>(a, s);
  else if (force_tt != 16 && bwd_smem<32>(L) <= LDS_BUDGET) launch_bwd<32>(a, s);
  else { DX_REQUIRE(bwd_smem<16>(L) <= LDS_BUDGET, "dx_upsample_bwd: L=%d too long for the LDS tiles", L); launch_bwd<16>(a, s); }
  DX_LAUNCH_CHECK("dx_upsample_bwd");
  return DX_OK;
}

int dx_upsample_sym_bwd(const float* dxs_in, const float* dsigma, const float* xs, const float* z, const float* dur, const int* lens,
                        const float* wd, const float* bd, const float* wr, float* dxs_out, float* dz, float* dwr, float* dbr,
                        int B, int L, int Dm, void* stream) {
  DX_REQUIRE(dxs_in && dsigma && xs && z && dur && lens && wd && bd && wr && dxs_out && dz && dwr && dbr, "dx_upsample_sym_bwd: null pointer");
  DX_REQUIRE(Dm == D && B > 0 && L > 0, "dx_upsample_sym_bwd: bad dims (D must be 128)");
  SymBwdArgs a{dxs_in, dsigma, xs, z, dur, lens, wd, bd, wr, dxs_out, dz, dwr, dbr, B, L};
  const long rows = (long)B * L;
  // every block ends in 129 atomics on the same 129 addresses (dwr, dbr): ~70 ns each when the blocks finish together, so the launch time
  // follows the block count on one side and the rows per wave on the other (1024 / 256 / 128 / 64 blocks: 40 / 20 / 24 / 37 us)
  static const int nblk_env = getenv("DX_SYM_BLOCKS") ? atoi(getenv("DX_SYM_BLOCKS")) : 256;
  hipLaunchKernelGGL(upsample_sym_bwd_kernel, dim3((int)std::min<long>((rows + 3) / 4, nblk_env)), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK("dx_upsample_sym_bwd");
  return DX_OK;
}

}  // extern "C"

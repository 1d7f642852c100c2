// Multi-head self-attention with key-padding mask and dropout on the probabilities, head_dim = 64, gfx950.
// Replaces nn.MultiheadAttention's slow path as the reference calls it (src/daft_exprt/model.py:165-186, :255):
// no (B*H, N, N) score matrix in HBM, no head-averaged weights (the caller discards them).
//
// Layout: qkv is the in-projection output, fp32 [B*N][3*D] (q | k | v, head h = columns h*64..h*64+63 of each third);
// ctx is [B*N][D].  lse = log-sum-exp of the scaled, masked scores per (b, h, query) is kept for the backward.
//
// Forward / dQ kernels: a workgroup = 64 queries of one (b, h), 4 waves x 16 queries, looping over 64-key tiles in LDS.
// Scores are computed TRANSPOSED, S^T = K Q^T (MFMA A = K rows from LDS, B = Q^T from registers), so a lane owns one
// query column and 4 consecutive keys per 16x16 tile: the softmax statistics are per-lane scalars (+2 shuffles), and the
// probability registers are already the MFMA B operand of O^T += V^T P^T (v_mfma_f32_16x16x4_f32, k = 4*lanegroup + reg):
// no LDS round trip for P.  dK/dV kernel: a workgroup = 64 keys, 4 waves x 16 keys, looping over 64-query tiles; there the
// scores are computed as S = Q K^T so that P / dS registers are the B operand of dV^T += dO^T P and dK^T += Q^T dS.
// All accumulation is fp32; exp / log are the fp32 libm forms.
#include "dx_common.h"
#include <algorithm>

namespace {

constexpr int HD = 64;       // head dim
constexpr int TLD = 68;      // LDS row stride (floats) of a [64][64] tile: 16-B aligned rows, column reads conflict free
constexpr float QSCALE = 0.125f;  // 1/sqrt(64), exact
// bf16 kernels keep scores in log2 units (v_exp_f32 is 2^x): the S operand is scaled by QSCALE * log2(e) once when it is
// loaded, and the saved log-sum-exp is base 2 (scratch between the bf16 forward and backward, never user-visible)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float QSCALE2 = QSCALE * LOG2E;

struct AttnArgs {
  const float* qkv; int ld;        // [B*N][ld]
  const int* lens;
  float* ctx; int ldc;             // [B*N][ldc]
  float* lse;                      // [B][H][N]
  int B, N, H, D;
  uint64_t seed; uint32_t thresh; float inv_keep;
  const uint64_t* seed_offset;     // optional device scalar added to `seed` (captured HIP graphs: a new dropout stream per replay)
  const int* order;                // optional [B]: utterance indices, longest first (dx_length_order): blockIdx.z -> utterance
  int xcd_map;                     // 1: workgroups renumbered so that the tiles of an (utterance, head) pair share an XCD (attn_block)
};

__device__ __forceinline__ f32x4 mma4(const float4& a, const float4& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  return c;
}

// stage rows [r0, r0+64) x 64 columns (starting at column col0) of a [B*N][ld] matrix into LDS; rows >= N are zero
__device__ __forceinline__ void stage_tile(float* dst, const float* base, int ld, int col0, int r0, int N, int tid) {
  for (int u = tid; u < 64 * 16; u += 256) {
    const int row = u >> 4, q = u & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < N) v = *reinterpret_cast<const float4*>(base + (size_t)(r0 + row) * ld + col0 + q * 4);
    *reinterpret_cast<float4*>(dst + row * TLD + q * 4) = v;
  }
}

// Workgroup -> (tile, head, utterance).  The tiles of one (utterance, head) pair all stream the same K / V (Q / dO) rows; dispatched in grid
// order (x fastest) consecutive tiles go to 8 DIFFERENT XCDs, so every XCD's L2 fetches the pair's operands for itself (PMC: 3.2-3.6x the
// algorithmic bytes, 3.7 GB of the step's ~16 GB).  Renumbered, XCD x runs the pairs p = 8 i + x with all their tiles back to back: one
// fetch per pair, L2 hits for the rest.  Needs (heads x utterances) % 8 == 0 (else the plain order).  Pure scheduling.
struct AttnBlock { int tile, h, z; };
__device__ __forceinline__ AttnBlock attn_block(int xcd_map) {
  AttnBlock o{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  if (xcd_map) {
    const int ntx = gridDim.x, H = gridDim.y;
    const int lin = blockIdx.x + ntx * (blockIdx.y + H * blockIdx.z);
    const int xcd = lin & 7, j = lin >> 3;
    const int pair = (j / ntx) * 8 + xcd;
    o.tile = j - (j / ntx) * ntx;
    o.h = pair % H;
    o.z = pair / H;
  }
  return o;
}

__device__ __forceinline__ uint64_t drop_index(int bh, int N, int q, int key) {
  return ((uint64_t)((size_t)bh * N + q) << 16) | (uint64_t)key;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs a_) {
  AttnArgs a = a_;
  if (a.seed_offset) a.seed += *a.seed_offset;
  __shared__ __attribute__((aligned(16))) float Ks[64 * TLD];
  __shared__ __attribute__((aligned(16))) float Vs[64 * TLD];
  const AttnBlock blk = attn_block(a.xcd_map);
  const int b = a.order ? a.order[blk.z] : blk.z, h = blk.h, q0 = blk.tile * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int len = a.lens[b];
  const float* base = a.qkv + (size_t)b * a.N * a.ld;
  const int qrow = q0 + wave * 16 + r;                 // this lane's query
  float* out = a.ctx + ((size_t)b * a.N + qrow) * a.ldc + h * HD;
  if (q0 >= len) {                                     // padded queries: defined zeros (they feed GEMMs later)
    if (qrow < a.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<float4*>(out + dt * 16 + g * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g == 0) a.lse[((size_t)b * a.H + h) * a.N + qrow] = 0.f;
    }
    return;
  }
  const int qload = min(qrow, a.N - 1);
  float4 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)qload * a.ld + h * HD + ks * 16 + g * 4);
    qf[ks] = make_float4(v.x * QSCALE, v.y * QSCALE, v.z * QSCALE, v.w * QSCALE);
  }
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int bh = b * a.H + h;
  const int ntiles = (len + 63) / 64;
  for (int kt0 = 0; kt0 < ntiles; ++kt0) {
    const int kbase = kt0 * 64;
    __syncthreads();
    stage_tile(Ks, base, a.ld, a.D + h * HD, kbase, a.N, tid);
    stage_tile(Vs, base, a.ld, 2 * a.D + h * HD, kbase, a.N, tid);
    __syncthreads();
    f32x4 st[4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float4 kf = *reinterpret_cast<const float4*>(Ks + (kt * 16 + r) * TLD + ks * 16 + g * 4);
        acc = mma4(kf, qf[ks], acc);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (kbase + kt * 16 + g * 4 + e >= len) acc[e] = -INFINITY;
        mx = fmaxf(mx, acc[e]);
      }
      st[kt] = acc;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);              // finite: key 0 is always valid
    const float alpha = expf(m_run - m_new);
    float ls = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      float keep[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.thresh) dx_dropout_scale4(a.seed, drop_index(bh, a.N, qrow, kbase + kt * 16 + g * 4), a.thresh, a.inv_keep, keep);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = expf(st[kt][e] - m_new);
        ls += p;
        st[kt][e] = p * keep[e];
      }
    }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    l_run = l_run * alpha + ls;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 acc = o[dt];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] *= alpha;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float vf = Vs[(kt * 16 + g * 4 + e) * TLD + dt * 16 + r];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, st[kt][e], acc, 0, 0, 0);
        }
      }
      o[dt] = acc;
    }
  }
  if (qrow < a.N) {
    const bool valid = qrow < len;
    const float inv = valid ? 1.f / l_run : 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<float4*>(out + dt * 16 + g * 4) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
    if (g == 0) a.lse[((size_t)b * a.H + h) * a.N + qrow] = valid ? m_run + logf(l_run) : 0.f;
  }
}

// delta[b,h,q] = sum_d dctx[q][h*64+d] * ctx[q][h*64+d]     (wave per (row, head))
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ dctx, const float* __restrict__ ctx, int ldc,
                                                         float* __restrict__ delta, int B, int N, int H) {
  const int lane = threadIdx.x & 63;
  const long items = (long)B * N * H;
  for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += (long)gridDim.x * 4) {
    const int h = (int)(it % H);
    const long row = it / H;
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    const float v = dctx[row * ldc + h * HD + lane] * ctx[row * ldc + h * HD + lane];
    const float s = dx_wave_sum(v);
    if (lane == 0) delta[((size_t)b * H + h) * N + n] = s;
  }
}

struct AttnBwdArgs {
  const float* qkv; int ld;
  const float* dctx; int ldc;
  const float* lse; const float* delta;   // [B][H][N]
  const int* lens;
  float* dqkv; int ldg;                   // [B*N][ldg], same column layout as qkv
  int B, N, H, D;
  uint64_t seed; uint32_t thresh; float inv_keep;
  const float* ctx;                       // [B*N][ldc]: the bf16 dQ kernel computes delta itself (and stores it for dK/dV)
  float* delta_out;
  const uint64_t* seed_offset;
  const int* order;                       // as in AttnArgs
  int xcd_map;
};

// ------------------------------------------------------------------------------------------------
// dQ: same geometry as the forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnBwdArgs a_) {
  AttnBwdArgs a = a_;
  if (a.seed_offset) a.seed += *a.seed_offset;
  __shared__ __attribute__((aligned(16))) float Ks[64 * TLD];
  __shared__ __attribute__((aligned(16))) float Vs[64 * TLD];
  const AttnBlock blk = attn_block(a.xcd_map);
  const int b = a.order ? a.order[blk.z] : blk.z, h = blk.h, q0 = blk.tile * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int len = a.lens[b];
  const float* base = a.qkv + (size_t)b * a.N * a.ld;
  const int qrow = q0 + wave * 16 + r;
  float* out = a.dqkv + ((size_t)b * a.N + qrow) * a.ldg + h * HD;
  if (q0 >= len) {
    if (qrow < a.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<float4*>(out + dt * 16 + g * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int qload = min(qrow, a.N - 1);
  const int bh = b * a.H + h;
  float4 qf[4], gf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)qload * a.ld + h * HD + ks * 16 + g * 4);
    qf[ks] = make_float4(v.x * QSCALE, v.y * QSCALE, v.z * QSCALE, v.w * QSCALE);
    gf[ks] = *reinterpret_cast<const float4*>(a.dctx + ((size_t)b * a.N + qload) * a.ldc + h * HD + ks * 16 + g * 4);
  }
  const float lse_q = a.lse[(size_t)bh * a.N + qload];
  const float delta_q = a.delta[(size_t)bh * a.N + qload];
  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (len + 63) / 64;
  for (int kt0 = 0; kt0 < ntiles; ++kt0) {
    const int kbase = kt0 * 64;
    __syncthreads();
    stage_tile(Ks, base, a.ld, a.D + h * HD, kbase, a.N, tid);
    stage_tile(Vs, base, a.ld, 2 * a.D + h * HD, kbase, a.N, tid);
    __syncthreads();
    f32x4 ds[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float4 kf = *reinterpret_cast<const float4*>(Ks + (kt * 16 + r) * TLD + ks * 16 + g * 4);
        const float4 vf = *reinterpret_cast<const float4*>(Vs + (kt * 16 + r) * TLD + ks * 16 + g * 4);
        s = mma4(kf, qf[ks], s);
        dp = mma4(vf, gf[ks], dp);
      }
      float keep[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.thresh) dx_dropout_scale4(a.seed, drop_index(bh, a.N, qrow, kbase + kt * 16 + g * 4), a.thresh, a.inv_keep, keep);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool masked = kbase + kt * 16 + g * 4 + e >= len;
        const float p = masked ? 0.f : expf(s[e] - lse_q);
        s[e] = p * (dp[e] * keep[e] - delta_q) * QSCALE;
      }
      ds[kt] = s;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 acc = dq[dt];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float kf = Ks[(kt * 16 + g * 4 + e) * TLD + dt * 16 + r];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, ds[kt][e], acc, 0, 0, 0);
        }
      }
      dq[dt] = acc;
    }
  }
  if (qrow < a.N) {
    const float z = qrow < len ? 1.f : 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<float4*>(out + dt * 16 + g * 4) = make_float4(dq[dt][0] * z, dq[dt][1] * z, dq[dt][2] * z, dq[dt][3] * z);
  }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: workgroup = 64 keys of one (b, h); wave = 16 keys; loop over 64-query tiles
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnBwdArgs a_) {
  AttnBwdArgs a = a_;
  if (a.seed_offset) a.seed += *a.seed_offset;
  __shared__ __attribute__((aligned(16))) float Qs[64 * TLD];
  __shared__ __attribute__((aligned(16))) float Gs[64 * TLD];
  __shared__ float lse_s[64], delta_s[64];
  const AttnBlock blk = attn_block(a.xcd_map);
  const int b = a.order ? a.order[blk.z] : blk.z, h = blk.h, k0 = blk.tile * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int len = a.lens[b];
  const float* base = a.qkv + (size_t)b * a.N * a.ld;
  const float* gbase = a.dctx + (size_t)b * a.N * a.ldc;
  const int krow = k0 + wave * 16 + r;                 // this lane's key
  float* outk = a.dqkv + ((size_t)b * a.N + krow) * a.ldg + a.D + h * HD;
  float* outv = a.dqkv + ((size_t)b * a.N + krow) * a.ldg + 2 * a.D + h * HD;
  if (k0 >= len) {
    if (krow < a.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<float4*>(outk + dt * 16 + g * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(outv + dt * 16 + g * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    return;
  }
  const int kload = min(krow, a.N - 1);
  const bool key_valid = krow < len;
  const int bh = b * a.H + h;
  float4 kf[4], vf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)kload * a.ld + a.D + h * HD + ks * 16 + g * 4);
    kf[ks] = make_float4(v.x * QSCALE, v.y * QSCALE, v.z * QSCALE, v.w * QSCALE);
    vf[ks] = *reinterpret_cast<const float4*>(base + (size_t)kload * a.ld + 2 * a.D + h * HD + ks * 16 + g * 4);
  }
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int ntiles = (len + 63) / 64;                  // queries >= len have zero upstream gradient
  for (int qt0 = 0; qt0 < ntiles; ++qt0) {
    const int qbase = qt0 * 64;
    __syncthreads();
    stage_tile(Qs, base, a.ld, h * HD, qbase, a.N, tid);
    stage_tile(Gs, gbase, a.ldc, h * HD, qbase, a.N, tid);
    if (tid < 64) {
      const int q = qbase + tid;
      lse_s[tid] = q < a.N ? a.lse[(size_t)bh * a.N + q] : 0.f;
      delta_s[tid] = q < a.N ? a.delta[(size_t)bh * a.N + q] : 0.f;
    }
    __syncthreads();
    f32x4 pd[4], ds[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float4 qf = *reinterpret_cast<const float4*>(Qs + (qt * 16 + r) * TLD + ks * 16 + g * 4);
        const float4 gf = *reinterpret_cast<const float4*>(Gs + (qt * 16 + r) * TLD + ks * 16 + g * 4);
        s = mma4(qf, kf[ks], s);
        dp = mma4(gf, vf[ks], dp);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ql = qt * 16 + g * 4 + e, q = qbase + ql;
        const bool live = key_valid && q < len;
        const float p = live ? expf(s[e] - lse_s[ql]) : 0.f;
        float keep = 1.f;
        if (a.thresh) keep = dx_dropout_scale(a.seed, drop_index(bh, a.N, q, krow), a.thresh, a.inv_keep);
        pd[qt][e] = p * keep;
        ds[qt][e] = p * (dp[e] * keep - delta_s[ql]) * QSCALE;
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 accv = dv[dt], acck = dk[dt];
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gcol = Gs[(qt * 16 + g * 4 + e) * TLD + dt * 16 + r];
          const float qcol = Qs[(qt * 16 + g * 4 + e) * TLD + dt * 16 + r];
          accv = __builtin_amdgcn_mfma_f32_16x16x4f32(gcol, pd[qt][e], accv, 0, 0, 0);
          acck = __builtin_amdgcn_mfma_f32_16x16x4f32(qcol, ds[qt][e], acck, 0, 0, 0);
        }
      }
      dv[dt] = accv; dk[dt] = acck;
    }
  }
  if (krow < a.N) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *reinterpret_cast<float4*>(outk + dt * 16 + g * 4) = make_float4(dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]);
      *reinterpret_cast<float4*>(outv + dt * 16 + g * 4) = make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
    }
  }
}


// =================================================================================================
// bf16 MFMA variants (bf16 operand mode): K/V (or Q/dO) tiles live in LDS as bf16 [64][64] with 128-byte rows and XOR-swizzled
// 16-byte slots.  Row-wise fragments are one ds_read_b128; the transposed fragments of V^T / K^T / dO^T / Q^T come from
// ds_read_b64_tr_b16 (a 16-lane group reads a 4-row x 16-column block and every lane receives one column), so two 16x16 fp32
// score tiles (32 keys) feed one v_mfma_f32_16x16x32_bf16 of the second product without any LDS round trip for P or dS.
// Softmax statistics, exp and accumulation stay fp32.
// =================================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int sw_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

#include "dx_rowvec.h"

// stage rows [r0, r0+64) x 64 fp32 columns (from col0) as bf16 into the swizzled image; rows >= N are zero
__device__ __forceinline__ void stage_tile_bf16(unsigned char* dst, const float* base, int ld, int col0, int r0, int N, int tid, float scale) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int u = tid + it * 256;
    const int row = u >> 4, q = u & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < N) v = *reinterpret_cast<const float4*>(base + (size_t)(r0 + row) * ld + col0 + q * 4);
    bf16x4 h;
    h[0] = (dx_h16)(v.x * scale); h[1] = (dx_h16)(v.y * scale); h[2] = (dx_h16)(v.z * scale); h[3] = (dx_h16)(v.w * scale);
    *reinterpret_cast<uint2*>(dst + sw_off(row, q >> 1) + ((q & 1) << 3)) = __builtin_bit_cast(uint2, h);
  }
}

// software pipeline: the NEXT tile's rows are fetched into registers (4 x 16 bytes per thread per tensor) before the current
// tile's MFMA/softmax work, and converted + written to LDS after it
#define DX_TILE_LOAD(QT_, REG, BASE, LD, COL0, R0, NROWS)                                                         \
  if constexpr (sizeof(QT_) == 4) {                                                                               \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                           \
      const int u_ = tid + it * 256;                                                                              \
      const int row_ = u_ >> 4, q_ = u_ & 15;                                                                     \
      f32x4 v_ = f32x4{0.f, 0.f, 0.f, 0.f};                                                                       \
      if ((R0) + row_ < (NROWS)) v_ = *reinterpret_cast<const f32x4*>((BASE) + (size_t)((R0) + row_) * (LD) + (COL0) + q_ * 4); \
      REG[it] = v_;                                                                                               \
    }                                                                                                             \
  } else {                                                                                                        \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                                           \
      const int u_ = tid + it * 256;                                                                              \
      const int row_ = u_ >> 3, q_ = u_ & 7;                                                                      \
      f32x4 v_ = f32x4{0.f, 0.f, 0.f, 0.f};                                                                       \
      if ((R0) + row_ < (NROWS)) v_ = *reinterpret_cast<const f32x4*>((BASE) + (size_t)((R0) + row_) * (LD) + (COL0) + q_ * 8); \
      REG[it] = v_;                                                                                               \
    }                                                                                                             \
  }
#define DX_TILE_STORE(QT_, REG, DST)                                                                              \
  if constexpr (sizeof(QT_) == 4) {                                                                               \
    _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                           \
      const int u_ = tid + it * 256;                                                                              \
      const int row_ = u_ >> 4, q_ = u_ & 15;                                                                     \
      bf16x4 h_;                                                                                                  \
      h_[0] = (dx_h16)REG[it][0]; h_[1] = (dx_h16)REG[it][1]; h_[2] = (dx_h16)REG[it][2]; h_[3] = (dx_h16)REG[it][3]; \
      *reinterpret_cast<uint2*>((DST) + sw_off(row_, q_ >> 1) + ((q_ & 1) << 3)) = __builtin_bit_cast(uint2, h_); \
    }                                                                                                             \
  } else {                                                                                                        \
    _Pragma("unroll") for (int it = 0; it < 2; ++it) {                                                           \
      const int u_ = tid + it * 256;                                                                              \
      *reinterpret_cast<f32x4*>((DST) + sw_off(u_ >> 3, u_ & 7)) = REG[it];                                       \
    }                                                                                                             \
  }

// row fragment: 8 consecutive columns (32*ks + 8*g ..) of row `row`
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* tile, int row, int ks, int g) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(tile + sw_off(row, ks * 4 + g)));
}

// transposed half fragment: rows row0..row0+3 (row0 already includes the lane group's 4*g), columns col0..col0+15;
// lane (lane & 15) receives column col0 + (lane & 15), rows row0..row0+3
__device__ __forceinline__ s16x4 tr_half(const unsigned char* tile, int row0, int col0, int lane) {
  const int li = lane & 15, q = li >> 2, p = li & 3;
  const unsigned char* addr = tile + sw_off(row0 + q, (col0 >> 3) + (p >> 1)) + ((p & 1) << 3);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr));
}
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* tile, int rowA, int rowB, int col0, int lane) {
  const s16x4 lo = tr_half(tile, rowA, col0, lane), hi = tr_half(tile, rowB, col0, lane);
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {
  bf16x8 h;
  h[0] = (dx_h16)a[0]; h[1] = (dx_h16)a[1]; h[2] = (dx_h16)a[2]; h[3] = (dx_h16)a[3];
  h[4] = (dx_h16)b[0]; h[5] = (dx_h16)b[1]; h[6] = (dx_h16)b[2]; h[7] = (dx_h16)b[3];
  return h;
}
// 8 consecutive fp32 values of a global row -> bf16x8 (scaled)
__device__ __forceinline__ bf16x8 load_row8(const float* p, float scale) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  bf16x8 h;
  h[0] = (dx_h16)(a.x * scale); h[1] = (dx_h16)(a.y * scale); h[2] = (dx_h16)(a.z * scale); h[3] = (dx_h16)(a.w * scale);
  h[4] = (dx_h16)(b.x * scale); h[5] = (dx_h16)(b.y * scale); h[6] = (dx_h16)(b.z * scale); h[7] = (dx_h16)(b.w * scale);
  return h;
}
__device__ __forceinline__ bf16x8 load_row8(const dx_h16* p, float scale) {
  bf16x8 h = *reinterpret_cast<const bf16x8*>(p);
  if (scale != 1.f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (dx_h16)((float)h[e] * scale);
  }
  return h;
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) { *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d); }
__device__ __forceinline__ void store4(dx_h16* p, float a, float b, float c, float d) {
  bf16x4 h; h[0] = (dx_h16)a; h[1] = (dx_h16)b; h[2] = (dx_h16)c; h[3] = (dx_h16)d;
  *reinterpret_cast<bf16x4*>(p) = h;
}
#define DX_MFMA_BF16(A, B, C) DX_MFMA_H16((A), (B), (C))
// Timing ablations (tools/ablation_build.py; results are then numerically wrong on purpose): DX_ATTN_ABL = 1: no exp2 (p = s);
// 2: key / query tiles are staged once and reused (no LDS stores, barriers or global loads in the loop); 3: the first products
// (S, dP) are skipped; 4: the second products (O, dV, dK, dQ) are skipped.
#ifndef DX_ATTN_ABL
#define DX_ATTN_ABL 0
#endif
#if DX_ATTN_ABL == 1
#define DX_EXP2(X) (X)
#else
#define DX_EXP2(X) __builtin_amdgcn_exp2f(X)
#endif
#if DX_ATTN_ABL == 3
#define DX_MFMA_1ST(A, B, C) (C)
#else
#define DX_MFMA_1ST(A, B, C) DX_MFMA_H16((A), (B), (C))
#endif
#if DX_ATTN_ABL == 4
#define DX_MFMA_2ND(A, B, C) (C)
#else
#define DX_MFMA_2ND(A, B, C) DX_MFMA_H16((A), (B), (C))
#endif
__device__ __forceinline__ float dx_max2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }   // no NaNs here: scores are finite or -inf
// (fmaxf() makes hipcc canonicalise every operand first - a v_max_f32 x, x, x each, 16 per key tile.  An inline-asm v_max3_f32 on the
// MFMA results avoids that but is WRONG: the compiler's hazard recogniser does not look inside inline asm, so the asm read the
// accumulators before the matrix pipe had written them - maxima of stale data, a 0.4 % output error that the eager-vs-graph
// bitwise test caught.  Inline asm may only consume values that an ordinary VALU instruction produced.)

template <typename QT, typename CT>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16_kernel(const AttnArgs a_) {
  AttnArgs a = a_;
  if (a.seed_offset) a.seed += *a.seed_offset;
  __shared__ __attribute__((aligned(16))) unsigned char Ks[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[64 * 128];
  const AttnBlock blk = attn_block(a.xcd_map);
  const int b = a.order ? a.order[blk.z] : blk.z, h = blk.h, q0 = blk.tile * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int len = a.lens[b];
  const QT* base = reinterpret_cast<const QT*>(a.qkv) + (size_t)b * a.N * a.ld;
  const int qrow = q0 + wave * 16 + r;
  CT* out = reinterpret_cast<CT*>(a.ctx) + ((size_t)b * a.N + qrow) * a.ldc + h * HD;
  if (q0 >= len) {
    if (qrow < a.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4(out + dt * 16 + g * 4, 0.f, 0.f, 0.f, 0.f);
      if (g == 0) a.lse[((size_t)b * a.H + h) * a.N + qrow] = 0.f;
    }
    return;
  }
  const int qload = min(qrow, a.N - 1);
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = load_row8(base + (size_t)qload * a.ld + h * HD + ks * 32 + g * 8, QSCALE2);
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int bh = b * a.H + h;
  const int ntiles = (len + 63) / 64;
  // dropout draw index of (query row, 4 consecutive keys) = drop_index(..) >> 2: the row part is fixed per lane
  const uint64_t drow = (uint64_t)((size_t)bh * a.N + qrow) << 14;
  const uint32_t thresh_v = a.thresh;
  f32x4 kreg[4], vreg[4];
  DX_TILE_LOAD(QT, kreg, base, a.ld, a.D + h * HD, 0, a.N)
  DX_TILE_LOAD(QT, vreg, base, a.ld, 2 * a.D + h * HD, 0, a.N)
  for (int kt0 = 0; kt0 < ntiles; ++kt0) {
    const int kbase = kt0 * 64;
    if (DX_ATTN_ABL != 2 || kt0 == 0) {
    __syncthreads();
    DX_TILE_STORE(QT, kreg, Ks)
    DX_TILE_STORE(QT, vreg, Vs)
    __syncthreads();
    if (kt0 + 1 < ntiles && DX_ATTN_ABL != 2) {
      DX_TILE_LOAD(QT, kreg, base, a.ld, a.D + h * HD, kbase + 64, a.N)
      DX_TILE_LOAD(QT, vreg, base, a.ld, 2 * a.D + h * HD, kbase + 64, a.N)
    }
    }
    f32x4 st[4];
    float mx = -INFINITY;
    const bool tail_tile = kbase + 64 > len;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) acc = DX_MFMA_1ST(row_frag(Ks, kt * 16 + r, ks, g), qf[ks], acc);
      if (tail_tile) {                                  // only the last key tile of a row can hold padding keys
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kbase + kt * 16 + g * 4 + e >= len) acc[e] = -INFINITY;
      }
      // max(a, b) as med3(a, b, +inf): ONE instruction on the accumulators.  fmaxf() makes hipcc canonicalise every operand first (a
      // v_max_f32 x, x each: 16 extra instructions per key tile in a VALU-bound loop); v_med3_f32 is not an IEEE maxnum and gets none.
      mx = dx_max2(mx, dx_max2(dx_max2(acc[0], acc[1]), dx_max2(acc[2], acc[3])));
      st[kt] = acc;
    }
    mx = dx_max2(mx, __shfl_xor(mx, 16, 64));
    mx = dx_max2(mx, __shfl_xor(mx, 32, 64));
    const float m_new = dx_max2(m_run, mx);
    // the running maxima settle after the first tiles: the rescale of O and l (an exp + 17 multiplies per lane) runs only in a
    // tile where some query of the wave saw a new maximum (wave-uniform branch)
    const bool rescale = __builtin_amdgcn_ballot_w64(m_new > m_run) != 0;
    float ls = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      float pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pk[e] = DX_EXP2(st[kt][e] - m_new);
        ls += pk[e];
      }
      // dropped probabilities become 0; the 1/(1-p) factor of the kept ones is applied once to O at the end
      if (a.thresh) dx_keep4(dx_rand64(a.seed, drow | (uint64_t)((kbase + kt * 16 + g * 4) >> 2)), thresh_v, pk, 0.f);   // pk: v_exp results
#pragma unroll
      for (int e = 0; e < 4; ++e) st[kt][e] = pk[e];
    }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    if (rescale) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[dt][e] *= alpha;
    }
    l_run += ls;
    m_run = m_new;
    const bf16x8 p01 = pack_pair(st[0], st[1]), p23 = pack_pair(st[2], st[3]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 acc = o[dt];
      acc = DX_MFMA_2ND(tr_pair(Vs, 0 + g * 4, 16 + g * 4, dt * 16, lane), p01, acc);
      acc = DX_MFMA_2ND(tr_pair(Vs, 32 + g * 4, 48 + g * 4, dt * 16, lane), p23, acc);
      o[dt] = acc;
    }
  }
  if (qrow < a.N) {
    const bool valid = qrow < len;
    const float inv = valid ? (a.thresh ? a.inv_keep : 1.f) / l_run : 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) store4(out + dt * 16 + g * 4, o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
    if (g == 0) a.lse[((size_t)b * a.H + h) * a.N + qrow] = valid ? m_run + __log2f(l_run) : 0.f;   // base 2 (see QSCALE2)
  }
}

// ------------------------------------------------------------------------------------------------
// Attention forward of BOTH heads + out-projection + dropout + residual + LayerNorm (+ FiLM + mask) of an FFT block in ONE launch
// (16-bit q/k/v and context, 2 heads x 64, model width 128).  As two launches (attn_fwd_bf16_kernel, then proj_ln_fwd_kernel of dx_rows.hip)
// the second one is a one-round, latency-bound shell of 17-19 us per frame-level block that re-reads the context; here a 512-thread
// workgroup owns a 64-query tile of one utterance: waves 0-3 run head 0 and waves 4-7 head 1 exactly as attn_fwd_bf16_kernel does (own
// K / V images, same arithmetic, same dropout draws), the normalised context goes to HBM (the backward needs it) AND into an LDS image,
// the 128 x 128 out-projection is 16 MFMAs per wave from that image, and the row pass of proj_ln_fwd_kernel runs on the result while other
// workgroups are still in their key loops (utterances differ in length, so the HBM-bound row pass hides under their VALU-bound loops).
// Results are bit-identical to the two launches (same operands, same K order, same row arithmetic): the test compares bitwise.
// ------------------------------------------------------------------------------------------------
constexpr float APL_LN_EPS = 1e-5f;          // (= LN_EPS of dx_rows.hip)
struct AttnProjLnArgs {
  AttnArgs at;
  const dx_h16* Wp; const float* proj_bias;                       // out-projection: forward pack of the (128, 128) weight, bias
  float* z; const float* res; const float* w; const float* bias;  // as LnArgs of dx_rows.hip (C = 128, fp32 rows)
  const float* film; int ld_film;
  float* y; dx_h16* y_h; float* mean; float* rstd;
  uint64_t seed_pre; uint32_t thresh_pre; float inv_keep_pre;
};

__global__ __launch_bounds__(512, 4) void attn_proj_ln_fwd_kernel(const AttnProjLnArgs p_) {
  typedef dx_h16 QT;
  typedef dx_h16 CT;
  constexpr int C = 128;
  AttnArgs a = p_.at;
  uint64_t seed_pre = p_.seed_pre;
  if (a.seed_offset) { const uint64_t o = *a.seed_offset; a.seed += o; seed_pre += o; }
  // LDS: K and V images of both heads (32 KB; after the key loop: the fp32 projection tile), the 16-bit context image (16 KB)
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 64 * 128 + 64 * 256];
  const int tid512 = threadIdx.x, lane = tid512 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid512 >> 6);
  const int h = wave >> 2, wq = wave & 3;
  const int tid = tid512 & 255;                         // the staging macros walk 256 threads per (head) image
  unsigned char* const Ks = smem + h * (64 * 128);
  unsigned char* const Vs = smem + 2 * 64 * 128 + h * (64 * 128);
  unsigned char* const Cs = smem + 4 * 64 * 128;        // [64 rows][256 B]: 16 slots of 8 channels, slot ^ (row & 15)
  float* const tile = reinterpret_cast<float*>(smem);   // [64][128] fp32, 4-float slots ^ (row & 15) (as proj_ln_fwd_kernel)
  const int r = lane & 15, g = lane >> 4;
  // workgroup -> (utterance, query tile): the tiles of an utterance run on ONE XCD (K / V fetched once), utterances go round the XCDs
  const int ntx = (a.N + 63) / 64;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int b = (j / ntx) * 8 + xcd, q0 = (j - (j / ntx) * ntx) * 64;
  if (b >= a.B) return;
  const int len = a.lens[b];
  const QT* base = reinterpret_cast<const QT*>(a.qkv) + (size_t)b * a.N * a.ld;
  const int qrow = q0 + wq * 16 + r;
  CT* out = reinterpret_cast<CT*>(a.ctx) + ((size_t)b * a.N + qrow) * a.ldc + h * HD;
  constexpr int E = RowVec<C>::E, LPR = RowVec<C>::LPR;
  const int l = lane % LPR, sub = lane / LPR;
  if (q0 >= len) {                                      // a padding tile: what the two launches leave there
    if (qrow < a.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4(out + dt * 16 + g * 4, 0.f, 0.f, 0.f, 0.f);
      if (g == 0) a.lse[((size_t)b * a.H + h) * a.N + qrow] = 0.f;
    }
    float zero[E];
#pragma unroll
    for (int e = 0; e < E; ++e) zero[e] = 0.f;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int n = q0 + wave * 8 + st * 4 + sub;
      if (n < a.N) {
        const long row = (long)b * a.N + n;
        row_store<C>(p_.z + row * C, l, zero);
        row_store<C>(p_.y + row * C, l, zero);
        if (p_.y_h) row_store<C>(p_.y_h + row * C, l, zero);
        if (l == 0) { p_.mean[row] = 0.f; p_.rstd[row] = 0.f; }
      }
    }
    return;
  }
  const int qload = min(qrow, a.N - 1);
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) qf[ks] = load_row8(base + (size_t)qload * a.ld + h * HD + ks * 32 + g * 8, QSCALE2);
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const int bh = b * a.H + h;
  const int ntiles = (len + 63) / 64;
  const uint64_t drow = (uint64_t)((size_t)bh * a.N + qrow) << 14;
  const uint32_t thresh_v = a.thresh;
  f32x4 kreg[4], vreg[4];
  DX_TILE_LOAD(QT, kreg, base, a.ld, a.D + h * HD, 0, a.N)
  DX_TILE_LOAD(QT, vreg, base, a.ld, 2 * a.D + h * HD, 0, a.N)
  for (int kt0 = 0; kt0 < ntiles; ++kt0) {
    const int kbase = kt0 * 64;
    __syncthreads();
    DX_TILE_STORE(QT, kreg, Ks)
    DX_TILE_STORE(QT, vreg, Vs)
    __syncthreads();
    if (kt0 + 1 < ntiles) {
      DX_TILE_LOAD(QT, kreg, base, a.ld, a.D + h * HD, kbase + 64, a.N)
      DX_TILE_LOAD(QT, vreg, base, a.ld, 2 * a.D + h * HD, kbase + 64, a.N)
    }
    f32x4 st[4];
    float mx = -INFINITY;
    const bool tail_tile = kbase + 64 > len;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) acc = DX_MFMA_1ST(row_frag(Ks, kt * 16 + r, ks, g), qf[ks], acc);
      if (tail_tile) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kbase + kt * 16 + g * 4 + e >= len) acc[e] = -INFINITY;
      }
      mx = dx_max2(mx, dx_max2(dx_max2(acc[0], acc[1]), dx_max2(acc[2], acc[3])));
      st[kt] = acc;
    }
    mx = dx_max2(mx, __shfl_xor(mx, 16, 64));
    mx = dx_max2(mx, __shfl_xor(mx, 32, 64));
    const float m_new = dx_max2(m_run, mx);
    const bool rescale = __builtin_amdgcn_ballot_w64(m_new > m_run) != 0;
    float ls = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      float pk[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pk[e] = DX_EXP2(st[kt][e] - m_new);
        ls += pk[e];
      }
      if (a.thresh) dx_keep4(dx_rand64(a.seed, drow | (uint64_t)((kbase + kt * 16 + g * 4) >> 2)), thresh_v, pk, 0.f);
#pragma unroll
      for (int e = 0; e < 4; ++e) st[kt][e] = pk[e];
    }
    ls += __shfl_xor(ls, 16, 64);
    ls += __shfl_xor(ls, 32, 64);
    if (rescale) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[dt][e] *= alpha;
    }
    l_run += ls;
    m_run = m_new;
    const bf16x8 p01 = pack_pair(st[0], st[1]), p23 = pack_pair(st[2], st[3]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 acc = o[dt];
      acc = DX_MFMA_2ND(tr_pair(Vs, 0 + g * 4, 16 + g * 4, dt * 16, lane), p01, acc);
      acc = DX_MFMA_2ND(tr_pair(Vs, 32 + g * 4, 48 + g * 4, dt * 16, lane), p23, acc);
      o[dt] = acc;
    }
  }
  // ---- context: to HBM as before, and (the very same 16-bit values) into the LDS image the projection reads ----------------------------
  {
    const bool valid = qrow < len;
    const float inv = valid ? (a.thresh ? a.inv_keep : 1.f) / l_run : 0.f;
    const int crow = wq * 16 + r;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 c4;
      c4[0] = (dx_h16)(o[dt][0] * inv); c4[1] = (dx_h16)(o[dt][1] * inv); c4[2] = (dx_h16)(o[dt][2] * inv); c4[3] = (dx_h16)(o[dt][3] * inv);
      if (qrow < a.N) *reinterpret_cast<bf16x4*>(out + dt * 16 + g * 4) = c4;
      const int slot = h * 8 + dt * 2 + (g >> 1);
      *reinterpret_cast<bf16x4*>(Cs + crow * 256 + ((slot ^ (crow & 15)) << 4) + ((g & 1) << 3)) = c4;
    }
    if (qrow < a.N && g == 0) a.lse[((size_t)b * a.H + h) * a.N + qrow] = valid ? m_run + __log2f(l_run) : 0.f;
  }
  // the row pass's global reads go out before the projection (a workgroup is one dependent chain)
  bool valid[2]; long rown[2];
  float rv[2][E];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int n = q0 + wave * 8 + st * 4 + sub;
    const bool inb = n < a.N;
    valid[st] = inb && n < len;
    rown[st] = inb ? (long)b * a.N + n : -1;
#pragma unroll
    for (int e = 0; e < E; ++e) rv[st][e] = 0.f;
    if (valid[st] && p_.res) row_load<C>(p_.res + rown[st] * C, l, rv[st]);
  }
  float wv[E], bv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { wv[e] = p_.w[row_col<C>(l, e)]; bv[e] = p_.bias[row_col<C>(l, e)]; }
  __syncthreads();                                      // the context image is complete; every K / V read has retired
  // ---- out-projection: wave = column tile (16 tokens) wq x four of the eight 16-channel row blocks; K order 0..3 as proj_ln_fwd_kernel --------
  {
    const int tok = wq * 16 + r;
    const bool live = q0 + tok < len;
    if (__builtin_amdgcn_ballot_w64(live) != 0) {
      bf16x8 xb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(Cs + tok * 256 + (((ks * 4 + g) ^ (tok & 15)) << 4));
      f32x4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = p_.proj_bias ? *reinterpret_cast<const f32x4*>(p_.proj_bias + (h * 4 + i) * 16 + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 wa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wa[i] = *reinterpret_cast<const bf16x8*>(p_.Wp + (size_t)((h * 4 + i) * 4 + ks) * 512 + lane * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = DX_MFMA_H16(wa[i], xb[ks], acc[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<f32x4*>(tile + tok * C + ((((h * 4 + i) * 4 + g) ^ (tok & 15)) << 2)) = acc[i];
    }
  }
  __syncthreads();
  // ---- row pass: exactly phase 2 of proj_ln_fwd_kernel, eight rows per wave in two steps of four -------------------------------------------
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int lrow = wave * 8 + st * 4 + sub;
    const long row = rown[st];
    const bool inb = row >= 0, vld = valid[st];
    float z[E];
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = 0.f;
    if (vld) {
#pragma unroll
      for (int k = 0; k < RowVec<C>::K; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tile + lrow * C + (((k * LPR + l) ^ (lrow & 15)) << 2));
        z[k * 4 + 0] = t[0]; z[k * 4 + 1] = t[1]; z[k * 4 + 2] = t[2]; z[k * 4 + 3] = t[3];
      }
      if (p_.thresh_pre) {
        float f[E];
        row_dropout<C>(seed_pre, (uint64_t)row, l, p_.thresh_pre, p_.inv_keep_pre, f);
#pragma unroll
        for (int e = 0; e < E; ++e) z[e] *= f[e];
      }
#pragma unroll
      for (int e = 0; e < E; ++e) z[e] += rv[st][e];
    }
    if (inb) row_store<C>(p_.z + row * C, l, z);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += z[e];
    const float mu = row_sum<C>(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { const float d = z[e] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(row_sum<C>(q) * (1.0f / C) + APL_LN_EPS);
    if (inb && l == 0) { p_.mean[row] = vld ? mu : 0.f; p_.rstd[row] = vld ? rs : 0.f; }
    float y[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int c = row_col<C>(l, e);
      float t = (z[e] - mu) * rs * wv[e] + bv[e];
      if (p_.film) t = p_.film[(size_t)b * p_.ld_film + c] * t + p_.film[(size_t)b * p_.ld_film + C + c];
      y[e] = vld ? t : 0.f;
    }
    if (inb) {
      row_store<C>(p_.y + row * C, l, y);
      if (p_.y_h) row_store<C>(p_.y_h + row * C, l, y);
    }
  }
}

// 8 consecutive values of a row as fp32
__device__ __forceinline__ void load8f(const float* p, float v[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8f(const dx_h16* p, float v[8]) {
  const bf16x8 h = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
}

template <typename QT, typename OT, typename CT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16_kernel(const AttnBwdArgs a_) {
  AttnBwdArgs a = a_;
  if (a.seed_offset) a.seed += *a.seed_offset;
  __shared__ __attribute__((aligned(16))) unsigned char Ks[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[64 * 128];
  const AttnBlock blk = attn_block(a.xcd_map);
  const int b = a.order ? a.order[blk.z] : blk.z, h = blk.h, q0 = blk.tile * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int len = a.lens[b];
  const QT* base = reinterpret_cast<const QT*>(a.qkv) + (size_t)b * a.N * a.ld;
  const int qrow = q0 + wave * 16 + r;
  OT* out = reinterpret_cast<OT*>(a.dqkv) + ((size_t)b * a.N + qrow) * a.ldg + h * HD;
  if (q0 >= len) {
    if (qrow < a.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) store4(out + dt * 16 + g * 4, 0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int qload = min(qrow, a.N - 1);
  const int bh = b * a.H + h;
  bf16x8 qf[2], gf[2];
  // delta[q] = sum_d dctx[q][d] * ctx[q][d]: the four lanes that share a query row hold its 64 head channels between them
  // (a separate launch for this cost 17 us per layer: 44 MB read again for 0.4 MB of output)
  float delta_q = 0.f;
  // the dO operand carries the 1/(1-p) of the dropped-out probabilities (dP = dO V^T is only ever used on kept elements)
  const float ik = a.thresh ? a.inv_keep : 1.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    qf[ks] = load_row8(base + (size_t)qload * a.ld + h * HD + ks * 32 + g * 8, QSCALE2);
    const size_t o = ((size_t)b * a.N + qload) * a.ldc + h * HD + ks * 32 + g * 8;
    float gv[8], cv[8];
    load8f(reinterpret_cast<const CT*>(a.dctx) + o, gv);           // dctx is stored like ctx (both fp32, or both 16-bit)
    load8f(reinterpret_cast<const CT*>(a.ctx) + o, cv);
    bf16x8 hg;
#pragma unroll
    for (int e = 0; e < 8; ++e) hg[e] = (dx_h16)(gv[e] * ik);
    gf[ks] = hg;
    delta_q += (gv[0] * cv[0] + gv[1] * cv[1]) + (gv[2] * cv[2] + gv[3] * cv[3]) + (gv[4] * cv[4] + gv[5] * cv[5]) + (gv[6] * cv[6] + gv[7] * cv[7]);
  }
  delta_q += __shfl_xor(delta_q, 16, 64);
  delta_q += __shfl_xor(delta_q, 32, 64);
  if (g == 0 && qrow < a.N) a.delta_out[(size_t)bh * a.N + qrow] = delta_q;
  const float lse_q = a.lse[(size_t)bh * a.N + qload];
  const float neg_delta_q = -delta_q;
  const f32x4 neg_lse4 = f32x4{-lse_q, -lse_q, -lse_q, -lse_q};
  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (len + 63) / 64;
  const uint64_t drow = (uint64_t)((size_t)bh * a.N + qrow) << 14;
  const uint32_t thresh_v = a.thresh;
  f32x4 kreg[4], vreg[4];
  DX_TILE_LOAD(QT, kreg, base, a.ld, a.D + h * HD, 0, a.N)
  DX_TILE_LOAD(QT, vreg, base, a.ld, 2 * a.D + h * HD, 0, a.N)
  for (int kt0 = 0; kt0 < ntiles; ++kt0) {
    const int kbase = kt0 * 64;
    __syncthreads();
    DX_TILE_STORE(QT, kreg, Ks)
    DX_TILE_STORE(QT, vreg, Vs)
    __syncthreads();
    if (kt0 + 1 < ntiles) {
      DX_TILE_LOAD(QT, kreg, base, a.ld, a.D + h * HD, kbase + 64, a.N)
      DX_TILE_LOAD(QT, vreg, base, a.ld, 2 * a.D + h * HD, kbase + 64, a.N)
    }
    f32x4 ds[4];
    const bool tail_tile = kbase + 64 > len;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      // S' = K Q^T - lse: the row constant is the INITIAL accumulator of the chain (a lane owns one query: the same four registers for
      // every key tile), so p = exp2(S') needs no subtraction (16 instructions per key tile in a VALU-bound loop)
      f32x4 s = neg_lse4, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        s = DX_MFMA_BF16(row_frag(Ks, kt * 16 + r, ks, g), qf[ks], s);
        dp = DX_MFMA_BF16(row_frag(Vs, kt * 16 + r, ks, g), gf[ks], dp);
      }
      // dP - delta on kept elements, -delta on dropped ones.  The subtraction comes first: the select is inline asm and must not be the
      // first reader of an MFMA result (dx_common.h, hazard rule)
      float u[4] = {dp[0] - delta_q, dp[1] - delta_q, dp[2] - delta_q, dp[3] - delta_q};
      if (a.thresh) dx_keep4(dx_rand64(a.seed, drow | (uint64_t)((kbase + kt * 16 + g * 4) >> 2)), thresh_v, u, neg_delta_q);
      float p[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = __builtin_amdgcn_exp2f(s[e]);
      if (tail_tile) {                                    // wave-uniform: only the last key tile can hold padding keys
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kbase + kt * 16 + g * 4 + e >= len) p[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = p[e] * u[e];                 // x QSCALE: applied once to dQ at the end
      ds[kt] = s;
    }
    const bf16x8 d01 = pack_pair(ds[0], ds[1]), d23 = pack_pair(ds[2], ds[3]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 acc = dq[dt];
      acc = DX_MFMA_BF16(tr_pair(Ks, 0 + g * 4, 16 + g * 4, dt * 16, lane), d01, acc);
      acc = DX_MFMA_BF16(tr_pair(Ks, 32 + g * 4, 48 + g * 4, dt * 16, lane), d23, acc);
      dq[dt] = acc;
    }
  }
  if (qrow < a.N) {
    const float z = qrow < len ? QSCALE : 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      store4(out + dt * 16 + g * 4, dq[dt][0] * z, dq[dt][1] * z, dq[dt][2] * z, dq[dt][3] * z);
  }
}

// three waves per SIMD where the prefetch registers hold 16-bit tiles (the model's configuration: 158 registers); the fp32-stored
// variants (kernel tests) would spill at that cap and stay at two
template <typename QT, typename OT, typename CT>
__global__ __launch_bounds__(256, (sizeof(QT) == 2 && sizeof(CT) == 2) ? 3 : 2) void attn_bwd_dkv_bf16_kernel(const AttnBwdArgs a_) {
  AttnBwdArgs a = a_;
  if (a.seed_offset) a.seed += *a.seed_offset;
  __shared__ __attribute__((aligned(16))) unsigned char Qs[64 * 128];
  __shared__ __attribute__((aligned(16))) unsigned char Gs[64 * 128];
  __shared__ __attribute__((aligned(16))) float lse_s[64], delta_s[64];
  const AttnBlock blk = attn_block(a.xcd_map);
  const int b = a.order ? a.order[blk.z] : blk.z, h = blk.h, k0 = blk.tile * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int len = a.lens[b];
  const QT* base = reinterpret_cast<const QT*>(a.qkv) + (size_t)b * a.N * a.ld;
  const CT* gbase = reinterpret_cast<const CT*>(a.dctx) + (size_t)b * a.N * a.ldc;
  const int krow = k0 + wave * 16 + r;
  OT* outk = reinterpret_cast<OT*>(a.dqkv) + ((size_t)b * a.N + krow) * a.ldg + a.D + h * HD;
  OT* outv = reinterpret_cast<OT*>(a.dqkv) + ((size_t)b * a.N + krow) * a.ldg + 2 * a.D + h * HD;
  if (k0 >= len) {
    if (krow < a.N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        store4(outk + dt * 16 + g * 4, 0.f, 0.f, 0.f, 0.f);
        store4(outv + dt * 16 + g * 4, 0.f, 0.f, 0.f, 0.f);
      }
    }
    return;
  }
  const int kload = min(krow, a.N - 1);
  const bool key_valid = krow < len;
  const int bh = b * a.H + h;
  bf16x8 kf[2], vf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    kf[ks] = load_row8(base + (size_t)kload * a.ld + a.D + h * HD + ks * 32 + g * 8, QSCALE2);
    vf[ks] = load_row8(base + (size_t)kload * a.ld + 2 * a.D + h * HD + ks * 32 + g * 8, a.thresh ? a.inv_keep : 1.f);   // carries 1/(1-p), see dQ
  }
  // this lane's 16-bit field of a draw (key krow & 3) as a v_perm_b32 selector over {hi, lo}: two bytes, zero-extended
  const uint32_t fsel = 0x0c0c0000u | (uint32_t)((2 * (krow & 3) + 1) << 8) | (uint32_t)(2 * (krow & 3));
  const uint32_t thresh_v = a.thresh;
  f32x4 dk[4], dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int ntiles = (len + 63) / 64;
  f32x4 qreg[4], greg[4];
  float lse_r = 0.f, delta_r = 0.f;
  DX_TILE_LOAD(QT, qreg, base, a.ld, h * HD, 0, a.N)
  DX_TILE_LOAD(CT, greg, gbase, a.ldc, h * HD, 0, a.N)
  if (tid < 64 && tid < a.N) { lse_r = a.lse[(size_t)bh * a.N + tid]; delta_r = a.delta[(size_t)bh * a.N + tid]; }
  for (int qt0 = 0; qt0 < ntiles; ++qt0) {
    const int qbase = qt0 * 64;
    if (DX_ATTN_ABL != 2 || qt0 == 0) {
    __syncthreads();
    DX_TILE_STORE(QT, qreg, Qs)
    DX_TILE_STORE(CT, greg, Gs)
    if (tid < 64) { lse_s[tid] = -lse_r; delta_s[tid] = delta_r; }     // (-lse: see the S' accumulator below)
    __syncthreads();
    }
    if (qt0 + 1 < ntiles && DX_ATTN_ABL != 2) {
      DX_TILE_LOAD(QT, qreg, base, a.ld, h * HD, qbase + 64, a.N)
      DX_TILE_LOAD(CT, greg, gbase, a.ldc, h * HD, qbase + 64, a.N)
      const int qn = qbase + 64 + tid;
      lse_r = (tid < 64 && qn < a.N) ? a.lse[(size_t)bh * a.N + qn] : 0.f;
      delta_r = (tid < 64 && qn < a.N) ? a.delta[(size_t)bh * a.N + qn] : 0.f;
    }
    const bool tail_q = qbase + 64 > len, tail_k = k0 + 64 > len;
    // Dropout words: one 64-bit draw covers 4 consecutive keys of a query row.  Here a lane owns ONE key and 16 query rows, and
    // the four lanes of a quad (keys 4m .. 4m+3) need the same 16 words: each lane draws 4 of them (rows g*4 + its quad index)
    // and the quad shares them by DPP instead of every lane drawing all 16 (the draws were ~40 % of this kernel's VALU time).
    // The tile is walked in two HALVES of 32 queries: scores / probabilities of a half (2 x 16 queries), then that half's eight
    // dV^T / dK^T MFMAs.  Only one half's P and dS registers are live at a time (16 instead of 32) and each draw is made where it is
    // used: 186 -> <= 168 registers, i.e. three waves per SIMD instead of two for a kernel that is bound by its dependent
    // chain (stage, barrier, multiply), not by instruction issue.  The accumulation order per element is unchanged (bitwise).
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 pd[2], ds[2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int qt = half * 2 + q2;
        uint32_t wlo = 0, whi = 0;
        if (a.thresh) {
          const uint64_t w = dx_rand64(a.seed, drop_index(bh, a.N, qbase + qt * 16 + g * 4 + (r & 3), krow) >> 2);
          wlo = (uint32_t)w; whi = (uint32_t)(w >> 32);
        }
        // S' = Q K^T - lse: -lse of the lane's four query rows (staged negated) is the initial accumulator, p = exp2(S') needs no subtraction
        f32x4 s = *reinterpret_cast<const f32x4*>(&lse_s[qt * 16 + g * 4]), dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          s = DX_MFMA_1ST(row_frag(Qs, qt * 16 + r, ks, g), kf[ks], s);
          dp = DX_MFMA_1ST(row_frag(Gs, qt * 16 + r, ks, g), vf[ks], dp);
        }
        float p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = DX_EXP2(s[e]);
        if (tail_k || tail_q) {                             // wave-uniform: padding keys / queries exist only in the last tiles
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!key_valid || qbase + qt * 16 + g * 4 + e >= len) p[e] = 0.f;
        }
        float pk[4] = {p[0], p[1], p[2], p[3]}, u[4], nd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { nd[e] = -delta_s[qt * 16 + g * 4 + e]; u[e] = dp[e] + nd[e]; }   // subtract BEFORE the asm select (hazard rule)
        if (a.thresh) {
          const uint32_t lo0 = __builtin_amdgcn_mov_dpp(wlo, 0x00, 0xF, 0xF, true), hi0 = __builtin_amdgcn_mov_dpp(whi, 0x00, 0xF, 0xF, true);
          const uint32_t lo1 = __builtin_amdgcn_mov_dpp(wlo, 0x55, 0xF, 0xF, true), hi1 = __builtin_amdgcn_mov_dpp(whi, 0x55, 0xF, 0xF, true);
          const uint32_t lo2 = __builtin_amdgcn_mov_dpp(wlo, 0xAA, 0xF, 0xF, true), hi2 = __builtin_amdgcn_mov_dpp(whi, 0xAA, 0xF, 0xF, true);
          const uint32_t lo3 = __builtin_amdgcn_mov_dpp(wlo, 0xFF, 0xF, 0xF, true), hi3 = __builtin_amdgcn_mov_dpp(whi, 0xFF, 0xF, 0xF, true);
          const uint32_t lo[4] = {lo0, lo1, lo2, lo3}, hi[4] = {hi0, hi1, hi2, hi3};
#pragma unroll
          for (int e = 0; e < 4; ++e)       // one byte-permute isolates this key's field; one compare drives both selects; 1/(1-p) rides on V and dV
            dx_keep2(__builtin_amdgcn_perm(hi[e], lo[e], fsel), thresh_v, pk[e], u[e], nd[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pd[q2][e] = pk[e];
          ds[q2][e] = p[e] * u[e];                                          // x QSCALE: applied once to dK at the end
        }
      }
      const bf16x8 pp = pack_pair(pd[0], pd[1]), dd = pack_pair(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = DX_MFMA_2ND(tr_pair(Gs, half * 32 + g * 4, half * 32 + 16 + g * 4, dt * 16, lane), pp, dv[dt]);
        dk[dt] = DX_MFMA_2ND(tr_pair(Qs, half * 32 + g * 4, half * 32 + 16 + g * 4, dt * 16, lane), dd, dk[dt]);
      }
    }
  }
  if (krow < a.N) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      store4(outk + dt * 16 + g * 4, dk[dt][0] * QSCALE, dk[dt][1] * QSCALE, dk[dt][2] * QSCALE, dk[dt][3] * QSCALE);
      const float ik = a.thresh ? a.inv_keep : 1.f;
      store4(outv + dt * 16 + g * 4, dv[dt][0] * ik, dv[dt][1] * ik, dv[dt][2] * ik, dv[dt][3] * ik);
    }
  }
}
#undef DX_MFMA_BF16
#undef DX_MFMA_1ST
#undef DX_MFMA_2ND
#undef DX_EXP2
#undef DX_TILE_LOAD
#undef DX_TILE_STORE

// order[rank] = b, rank = number of utterances that are longer than b (ties: lower index first): the dispatcher hands out workgroups in
// blockIdx order, so with blockIdx.z -> order[blockIdx.z] the longest utterances' workgroups start first.
__global__ __launch_bounds__(1024) void length_order_kernel(const int* __restrict__ lens, int* __restrict__ order, int B) {
  __shared__ int ls[1024];
  const int t = threadIdx.x;
  if (t < B) ls[t] = lens[t];
  __syncthreads();
  if (t >= B) return;
  const int mine = ls[t];
  int rank = 0;
  for (int j = 0; j < B; ++j) rank += (ls[j] > mine) || (ls[j] == mine && j < t);
  order[rank] = t;
}

int check_common(const char* who, const void* qkv, int ld, int B, int N, int H, int D) {
  DX_REQUIRE(qkv != nullptr, "%s: null pointer", who);
  DX_REQUIRE(B > 0 && N > 0 && H > 0 && D == H * HD, "%s: head dim must be 64 (D=%d, H=%d)", who, D, H);
  DX_REQUIRE(ld >= 3 * D && (ld % 4) == 0 && ((uintptr_t)qkv % 16) == 0, "%s: qkv must be 16-byte aligned with ld >= 3*D, ld %% 4 == 0", who);
  DX_REQUIRE(N < 65536, "%s: sequence length %d too long", who, N);
  return DX_OK;
}

}  // namespace

extern "C" {

// order[0..B) = the utterance indices sorted by length, longest first (stable).  Attention work per workgroup grows with the
// utterance's length (a 64-query workgroup walks every key tile of its utterance); workgroups are dispatched in blockIdx order, so
// handing the longest utterances out FIRST keeps a 14-tile workgroup from starting in the last round of a launch and setting its
// duration alone (C2: a third of the launch time was that tail).
int dx_length_order(const int* lens, int* order, int B, void* stream) {
  DX_REQUIRE(lens && order && B > 0 && B <= 1024, "dx_length_order: bad arguments (B <= 1024)");
  hipLaunchKernelGGL(length_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lens, order, B);
  DX_LAUNCH_CHECK("dx_length_order");
  return DX_OK;
}

int dx_attention_fwd(const void* qkvv, int ld, const int* lens, void* ctxv, int ldc, float* lse,
                     int B, int N, int H, int D, uint64_t seed, const uint64_t* seed_offset, float p_drop, int bf16, int qkv_bf16, int ctx_bf16,
                     const int* order, void* stream) {
  float* ctx = (float*)ctxv;
  const float* qkv = (const float*)qkvv;
  if (int rc = check_common("dx_attention_fwd", qkv, ld, B, N, H, D)) return rc;
  DX_REQUIRE(!qkv_bf16 || (bf16 && (ld % 8) == 0), "dx_attention_fwd: bf16-stored qkv needs bf16 mode and ld %% 8 == 0");
  DX_REQUIRE(lens && ctx && lse && ldc >= D && (ldc % 4) == 0 && ((uintptr_t)ctx % 16) == 0, "dx_attention_fwd: bad output arguments");
  DX_REQUIRE(!ctx_bf16 || (bf16 && (ldc % 8) == 0), "dx_attention_fwd: a 16-bit context needs the 16-bit operand mode and ldc %% 8 == 0");
  DX_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "dx_attention_fwd: dropout p out of range");
  AttnArgs a{qkv, ld, lens, ctx, ldc, lse, B, N, H, D, seed, (uint32_t)lrintf(p_drop * 65536.f), 1.f / (1.f - p_drop), seed_offset, order, 0};
  static const int xmap_env = getenv("DX_ATTN_XCD") ? atoi(getenv("DX_ATTN_XCD")) : 1;
  a.xcd_map = xmap_env && ((H * B) % 8 == 0);
  hipStream_t s = (hipStream_t)stream;
  dx_prof_begin(DX_PROF_ATTN_FWD, s);
  const dim3 grid(dx_cdiv(N, 64), H, B);
  if (bf16 && qkv_bf16 && ctx_bf16) hipLaunchKernelGGL((attn_fwd_bf16_kernel<dx_h16, dx_h16>), grid, dim3(256), 0, s, a);
  else if (bf16 && qkv_bf16) hipLaunchKernelGGL((attn_fwd_bf16_kernel<dx_h16, float>), grid, dim3(256), 0, s, a);
  else if (bf16 && ctx_bf16) hipLaunchKernelGGL((attn_fwd_bf16_kernel<float, dx_h16>), grid, dim3(256), 0, s, a);
  else if (bf16) hipLaunchKernelGGL((attn_fwd_bf16_kernel<float, float>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_fwd_kernel, dim3(dx_cdiv(N, 64), H, B), dim3(256), 0, s, a);
  dx_prof_end(DX_PROF_ATTN_FWD, s);
  DX_LAUNCH_CHECK("dx_attention_fwd");
  return DX_OK;
}

// dqkv (all three thirds, every row) from dctx; `delta` is scratch [B][H][N]
// dx_attention_fwd (16-bit q/k/v and context, 2 heads, D = 128) followed by dx_proj_ln_fwd of dx_rows.hip on its result, in ONE launch with
// bit-identical outputs: ctx / lse as dx_attention_fwd writes them; z = dropout(ctx W^T + proj_bias) + res, y = mask(FiLM(LayerNorm(z))), mean,
// rstd and the optional 16-bit copy of y as dx_proj_ln_fwd writes them (halo 0).  Replaces model.py:165-191 per FFT block (forward).
int dx_attention_proj_ln_fwd(const void* qkv, int ld, const int* lens, void* ctx, int ldc, float* lse, int B, int N, int H, int D,
                             uint64_t seed, const uint64_t* seed_offset, float p_drop,
                             const void* Wpack, const float* proj_bias, float* z, const float* res, const float* w, const float* bias,
                             const float* film, int ld_film, float* y, float* mean, float* rstd, uint64_t seed_pre, float p_pre,
                             void* y_bf16_copy, void* stream) {
  if (int rc = check_common("dx_attention_proj_ln_fwd", qkv, ld, B, N, H, D)) return rc;
  DX_REQUIRE(H == 2 && D == 128 && (ld % 8) == 0 && (ldc % 8) == 0 && ldc >= D, "dx_attention_proj_ln_fwd: 2 heads x 64, 16-bit rows (ld, ldc multiples of 8)");
  DX_REQUIRE(lens && ctx && lse && Wpack && z && w && bias && y && mean && rstd, "dx_attention_proj_ln_fwd: null pointer");
  DX_REQUIRE(((uintptr_t)ctx % 16) == 0 && ((uintptr_t)Wpack % 16) == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
             (!res || ((uintptr_t)res % 16) == 0), "dx_attention_proj_ln_fwd: pointers must be 16-byte aligned");
  DX_REQUIRE(p_drop >= 0.f && p_drop < 1.f && p_pre >= 0.f && p_pre < 1.f, "dx_attention_proj_ln_fwd: dropout p out of range");
  DX_REQUIRE(!film || ld_film >= 256, "dx_attention_proj_ln_fwd: ld_film too small");
  AttnProjLnArgs k{};
  k.at = AttnArgs{(const float*)qkv, ld, lens, (float*)ctx, ldc, lse, B, N, H, D, seed, (uint32_t)lrintf(p_drop * 65536.f), 1.f / (1.f - p_drop), seed_offset, nullptr, 0};
  k.Wp = (const dx_h16*)Wpack; k.proj_bias = proj_bias; k.z = z; k.res = res; k.w = w; k.bias = bias; k.film = film; k.ld_film = ld_film;
  k.y = y; k.y_h = (dx_h16*)y_bf16_copy; k.mean = mean; k.rstd = rstd;
  k.seed_pre = seed_pre; k.thresh_pre = (uint32_t)lrintf(p_pre * 65536.f); k.inv_keep_pre = 1.f / (1.f - p_pre);
  hipStream_t s = (hipStream_t)stream;
  dx_prof_begin(DX_PROF_ATTN_FWD, s);
  hipLaunchKernelGGL(attn_proj_ln_fwd_kernel, dim3(8 * dx_cdiv(N, 64) * dx_cdiv(B, 8)), dim3(512), 0, s, k);
  dx_prof_end(DX_PROF_ATTN_FWD, s);
  DX_LAUNCH_CHECK("dx_attention_proj_ln_fwd");
  return DX_OK;
}

int dx_attention_bwd(const void* qkvv, int ld, const void* ctxv, const void* dctxv, int ldc, const float* lse, float* delta,
                     const int* lens, void* dqkvv, int ldg, int B, int N, int H, int D, uint64_t seed, const uint64_t* seed_offset, float p_drop, int bf16,
                     int qkv_bf16, int dqkv_bf16, int ctx_bf16, const int* order, void* stream) {
  const float* qkv = (const float*)qkvv; float* dqkv = (float*)dqkvv; const float* ctx = (const float*)ctxv; const float* dctx = (const float*)dctxv;
  DX_REQUIRE(!ctx_bf16 || (bf16 && (ldc % 8) == 0), "dx_attention_bwd: a 16-bit context needs the 16-bit operand mode and ldc %% 8 == 0");
  if (int rc = check_common("dx_attention_bwd", qkv, ld, B, N, H, D)) return rc;
  DX_REQUIRE(!(qkv_bf16 || dqkv_bf16) || (bf16 && (ld % 8) == 0 && (ldg % 8) == 0), "dx_attention_bwd: bf16-stored qkv/dqkv need bf16 mode and ld %% 8 == 0");
  DX_REQUIRE(ctx && dctx && lse && delta && lens && dqkv, "dx_attention_bwd: null pointer");
  DX_REQUIRE(ldc >= D && (ldc % 4) == 0 && ldg >= 3 * D && (ldg % 4) == 0, "dx_attention_bwd: bad leading dimensions");
  DX_REQUIRE(((uintptr_t)dctx % 16) == 0 && ((uintptr_t)dqkv % 16) == 0, "dx_attention_bwd: pointers must be 16-byte aligned");
  DX_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "dx_attention_bwd: dropout p out of range");
  hipStream_t s = (hipStream_t)stream;
  const long items = (long)B * N * H;
  if (!bf16)                                       // the bf16 dQ kernel computes delta on the fly and leaves it for dK/dV
    hipLaunchKernelGGL(attn_delta_kernel, dim3((int)std::min<long>((items + 3) / 4, 8192)), dim3(256), 0, s, dctx, ctx, ldc, delta, B, N, H);
  AttnBwdArgs a{qkv, ld, dctx, ldc, lse, delta, lens, dqkv, ldg, B, N, H, D, seed, (uint32_t)lrintf(p_drop * 65536.f), 1.f / (1.f - p_drop), ctx, delta, seed_offset, order, 0};
  static const int xmap_env = getenv("DX_ATTN_XCD") ? atoi(getenv("DX_ATTN_XCD")) : 1;
  a.xcd_map = xmap_env && ((H * B) % 8 == 0);
  dx_prof_begin(DX_PROF_ATTN_BWD, s);
  if (bf16) {
    const dim3 grid(dx_cdiv(N, 64), H, B);
#define DX_ATTN_BWD(QT_, OT_)                                                                          \
    if (ctx_bf16) {                                                                                    \
      hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<QT_, OT_, dx_h16>), grid, dim3(256), 0, s, a);       \
      hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<QT_, OT_, dx_h16>), grid, dim3(256), 0, s, a);      \
    } else {                                                                                           \
      hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<QT_, OT_, float>), grid, dim3(256), 0, s, a);        \
      hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<QT_, OT_, float>), grid, dim3(256), 0, s, a);       \
    }
    if (qkv_bf16 && dqkv_bf16) { DX_ATTN_BWD(dx_h16, dx_h16) }
    else if (qkv_bf16) { DX_ATTN_BWD(dx_h16, float) }
    else if (dqkv_bf16) { DX_ATTN_BWD(float, dx_h16) }
    else { DX_ATTN_BWD(float, float) }
#undef DX_ATTN_BWD
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(dx_cdiv(N, 64), H, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(dx_cdiv(N, 64), H, B), dim3(256), 0, s, a);
  }
  dx_prof_end(DX_PROF_ATTN_BWD, s);
  DX_LAUNCH_CHECK("dx_attention_bwd");
  return DX_OK;
}

}  // extern "C"

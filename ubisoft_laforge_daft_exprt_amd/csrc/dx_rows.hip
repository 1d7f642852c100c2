// Memory-bound row kernels (one wave64 per token row, 16-byte / 8-byte coalesced accesses):
//   * residual + dropout + LayerNorm + FiLM + padding mask, forward and backward
//       (reference: model.py:188-191, :225-233, :256-258, prenet LayerNorms :655-669)
//   * symbol embedding + sinusoidal position + mask (model.py:597-604), decoder input (model.py:554-557)
//   * accent-encoder input sum: prenet + Conv1d(1->D) energy + Conv1d(1->D) pitch + position, masked (model.py:687-706)
//   * masked mean pooling (model.py:714), batched transposes, row L2-normalise (model.py:904), cross entropy (loss.py:85)
// All HBM-bound: the roofline for each is bytes moved / 8 TB/s (see DESIGN.md for the per-kernel byte counts).
#include "dx_common.h"
#include <cstdlib>
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr float LN_EPS = 1e-5f;

// ------------------------------------------------------------------------------------------------
// LayerNorm family.  z = drop_pre(a) + res  (written back over `a`), y = mask(film(drop_post(LN(z))))
// ------------------------------------------------------------------------------------------------
struct LnArgs {
  float* a;            // [rows][C] in: GEMM output; out: z (kept for backward)
  const float* res;    // [rows][C] or null
  const float* w; const float* bias;   // LN affine [C]
  const float* film; int ld_film;      // [B][>=2C] gamma | beta, or null
  const int* lens;     // [B] or null (no mask)
  int halo;            // rows n >= lens[b] + halo are padding nobody reads: zero-filled, not computed (halo 0 = the reference's mask)
  float* y;            // [rows][C]
  dx_h16* y_h;         // optional bf16 copy of y (GEMM operand of the next layer: same rounding as its staging would apply)
  float* mean; float* rstd;            // [rows]
  int B, N;
  uint64_t seed_pre; uint32_t thresh_pre; float inv_keep_pre;     // dropout on `a` (thresh 0 = off)
  uint64_t seed_post; uint32_t thresh_post; float inv_keep_post;  // dropout on LN output
  const uint64_t* seed_offset;                                    // optional device scalar added to both seeds (captured HIP graphs)
};

#include "dx_rowvec.h"

template <int C, typename IO>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnArgs a_) {
  LnArgs a = a_;
  if (a.seed_offset) { const uint64_t o = *a.seed_offset; a.seed_pre += o; a.seed_post += o; }
  constexpr int E = RowVec<C>::E, LPR = RowVec<C>::LPR, RPW = RowVec<C>::RPW;
  IO* const A = reinterpret_cast<IO*>(a.a);
  const IO* const R = reinterpret_cast<const IO*>(a.res);
  IO* const Y = reinterpret_cast<IO*>(a.y);
  const int lane = threadIdx.x & 63, l = lane % LPR, sub = lane / LPR;
  const long rows = (long)a.B * a.N;
  float wv[E], bv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { wv[e] = a.w[row_col<C>(l, e)]; bv[e] = a.bias[row_col<C>(l, e)]; }
  // the loop bound is wave-uniform; lanes whose row is out of range or padded run the arithmetic on zeros
  for (long base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW; base < rows; base += (long)gridDim.x * 4 * RPW) {
    const long row = base + sub;
    const bool inb = row < rows;
    const int b = inb ? (int)(row / a.N) : 0, n = inb ? (int)(row - (long)b * a.N) : 0;
    const bool valid = inb && (!a.lens || n < a.lens[b] + a.halo);
    float z[E];
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = 0.f;
    if (valid) {
      row_load<C>(A + row * C, l, z);
      if (a.thresh_pre) {
        float f[E];
        row_dropout<C>(a.seed_pre, (uint64_t)row, l, a.thresh_pre, a.inv_keep_pre, f);
#pragma unroll
        for (int e = 0; e < E; ++e) z[e] *= f[e];
      }
      if (a.res) {
        float rv[E];
        row_load<C>(R + row * C, l, rv);
#pragma unroll
        for (int e = 0; e < E; ++e) z[e] += rv[e];
      }
    }
    // padded rows: the output is zero by definition and nothing reads z / mean / rstd of them (the backward skips them too)
    if (inb && (a.thresh_pre || a.res)) row_store<C>(A + row * C, l, z);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += z[e];
    const float mu = row_sum<C>(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { const float d = z[e] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(row_sum<C>(q) * (1.0f / C) + LN_EPS);
    if (inb && l == 0) { a.mean[row] = valid ? mu : 0.f; a.rstd[row] = valid ? rs : 0.f; }
    float y[E], fpost[E];
    if (a.thresh_post) row_dropout<C>(a.seed_post, (uint64_t)row, l, a.thresh_post, a.inv_keep_post, fpost);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int c = row_col<C>(l, e);
      float t = (z[e] - mu) * rs * wv[e] + bv[e];
      if (a.thresh_post) t *= fpost[e];
      if (a.film) t = a.film[(size_t)b * a.ld_film + c] * t + a.film[(size_t)b * a.ld_film + C + c];
      y[e] = valid ? t : 0.f;
    }
    if (inb) {
      row_store<C>(Y + row * C, l, y);
      if constexpr (sizeof(IO) == 4 && C < 256) { if (a.y_h) row_store<C>(a.y_h + row * C, l, y); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Out-projection + dropout + residual + LayerNorm (+ FiLM + mask) in ONE launch (16-bit operand modes):
//   a = X W^T + proj_bias   (X: 16-bit [rows][128], W: 128 x 128 as the fragment-major forward pack)   then exactly ln_fwd_kernel<128>.
// A 64-row tile owns whole rows of the 128-wide output, so the GEMM result goes through LDS (XOR-swizzled 16-byte slots) straight
// into the row-wise LayerNorm pass: the projection output never makes its HBM round trip (512 B/row written + read) and one
// launch of the pair disappears.  Replaces attn.out_proj (model.py:165-186) + model.py:188-191 per FFT block.
// ------------------------------------------------------------------------------------------------
struct ProjLnArgs {
  const dx_h16* X; int ldx;
  const dx_h16* Wp;            // fragment-major pack of the 128 x 128 weight (dx_pack_weights forward image)
  const float* proj_bias;      // [128] or null
  LnArgs ln;                   // ln.a receives z; ln.y / y_h / mean / rstd as in ln_fwd
};

__global__ __launch_bounds__(256, 3) void proj_ln_fwd_kernel(const ProjLnArgs p_) {
  constexpr int C = 128;
  LnArgs a = p_.ln;
  if (a.seed_offset) { const uint64_t o = *a.seed_offset; a.seed_pre += o; a.seed_post += o; }
  constexpr int E = RowVec<C>::E, LPR = RowVec<C>::LPR;
  __shared__ __attribute__((aligned(16))) float tile[64 * C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int l = lane % LPR, sub = lane / LPR;
  const long rows = (long)a.B * a.N;
  const long row0 = (long)blockIdx.x * 64;
  // Everything the row pass needs from global memory is requested BEFORE the matrix phase (the residual rows of the lane's four steps,
  // the LayerNorm affine): a workgroup is one short dependent chain, and with these loads behind the barrier the 5.8 k-row
  // symbol-level launch took 24 us for 3 MB.
  bool valid[4]; int bidx[4]; long rown[4];
  float rv[4][E];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const long row = row0 + wave * 16 + st * 4 + sub;
    const bool inb = row < rows;
    const int b = inb ? (int)(row / a.N) : 0, n = inb ? (int)(row - (long)b * a.N) : 0;
    valid[st] = inb && (!a.lens || n < a.lens[b] + a.halo);
    bidx[st] = b;
    rown[st] = inb ? row : -1;
#pragma unroll
    for (int e = 0; e < E; ++e) rv[st][e] = 0.f;
    if (valid[st] && a.res) row_load<C>(a.res + row * C, l, rv[st]);
  }
  float wv[E], bv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { wv[e] = a.w[row_col<C>(l, e)]; bv[e] = a.bias[row_col<C>(l, e)]; }
  // ---- phase 1: the wave's 16 tokens x 128 output channels (W = A operand, X^T = B operand: a lane owns 4 channels of one token)
  {
    const long trow = row0 + wave * 16 + r;
    const long xrow = trow < rows ? trow : rows - 1;
    bool live = false;
    if (trow < rows) {
      const int b = (int)(trow / a.N), n = (int)(trow - (long)b * a.N);
      live = !a.lens || n < a.lens[b] + a.halo;
    }
    if (__builtin_amdgcn_ballot_w64(live) != 0) {           // a wave whose 16 rows are all padding skips the matrix work (phase 2 zero-fills)
      bf16x8 xb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(p_.X + xrow * p_.ldx + ks * 32 + g * 8);
      f32x4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i] = p_.proj_bias ? *reinterpret_cast<const f32x4*>(p_.proj_bias + i * 16 + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 wa[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wa[i] = *reinterpret_cast<const bf16x8*>(p_.Wp + (size_t)(i * 4 + ks) * 512 + lane * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = DX_MFMA_H16(wa[i], xb[ks], acc[i]);
      }
      const int tok = wave * 16 + r;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<f32x4*>(tile + tok * C + (((i * 4 + g) ^ (tok & 15)) << 2)) = acc[i];
    }
  }
  __syncthreads();
  // ---- phase 2: ln_fwd_kernel<128> on the tile (4 rows per wave-instruction, 16 lanes x 2 float4 per row), the four steps unrolled
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int lrow = wave * 16 + st * 4 + sub;
    const long row = rown[st];
    const bool inb = row >= 0, vld = valid[st];
    const int b = bidx[st];
    float z[E];
#pragma unroll
    for (int e = 0; e < E; ++e) z[e] = 0.f;
    if (vld) {
#pragma unroll
      for (int k = 0; k < RowVec<C>::K; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tile + lrow * C + (((k * LPR + l) ^ (lrow & 15)) << 2));
        z[k * 4 + 0] = t[0]; z[k * 4 + 1] = t[1]; z[k * 4 + 2] = t[2]; z[k * 4 + 3] = t[3];
      }
      if (a.thresh_pre) {
        float f[E];
        row_dropout<C>(a.seed_pre, (uint64_t)row, l, a.thresh_pre, a.inv_keep_pre, f);
#pragma unroll
        for (int e = 0; e < E; ++e) z[e] *= f[e];
      }
#pragma unroll
      for (int e = 0; e < E; ++e) z[e] += rv[st][e];
    }
    if (inb) row_store<C>(a.a + row * C, l, z);            // z, kept for the backward (zeros on padded rows)
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) s += z[e];
    const float mu = row_sum<C>(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) { const float d = z[e] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(row_sum<C>(q) * (1.0f / C) + LN_EPS);
    if (inb && l == 0) { a.mean[row] = vld ? mu : 0.f; a.rstd[row] = vld ? rs : 0.f; }
    float y[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int c = row_col<C>(l, e);
      float t = (z[e] - mu) * rs * wv[e] + bv[e];
      if (a.film) t = a.film[(size_t)b * a.ld_film + c] * t + a.film[(size_t)b * a.ld_film + C + c];
      y[e] = vld ? t : 0.f;
    }
    if (inb) {
      row_store<C>(a.y + row * C, l, y);
      if (a.y_h) row_store<C>(a.y_h + row * C, l, y);
    }
  }
}

struct LnBwdArgs {
  const float* dy;     // [rows][C]
  const float* z; const float* mean; const float* rstd;
  const float* w; const float* bias;
  const float* film; int ld_film;
  const int* lens; int halo;
  float* dz;           // [rows][C] gradient w.r.t. z (== residual branch gradient)
  float* da;           // [rows][C] gradient w.r.t. the pre-dropout GEMM output, or null (then dz serves)
  dx_h16* dg_h;        // optional bf16 copy of the gradient that feeds the GEMM backward (da if present, else dz)
  float* dw; float* dbias;             // [C] accumulated (atomics)
  float* dfilm; int ld_dfilm;          // [B][>=2C] accumulated, or null
  int B, N, rows_per_block;
  int relu_mask;       // multiply dz by (z > 0): ReLU sits right before the LayerNorm (prenet)
  uint64_t seed_pre; uint32_t thresh_pre; float inv_keep_pre;
  uint64_t seed_post; uint32_t thresh_post; float inv_keep_post;
  const uint64_t* seed_offset;
};

// grid: (blocks per batch row, B); each block walks rows of ONE batch row so FiLM gradients reduce per b.
// WAVES = waves per block (4 or 16): every block ends in one atomic per channel on the SAME 2 C addresses (dw, dbias), all blocks finish
// together, and same-address atomics serialise (~15 ns each: 672 four-wave blocks = a 10 us tail on a 25 us kernel).  Sixteen-wave
// blocks keep the same number of waves in flight with a quarter of the atomic chain (the sums fold through LDS first).
template <int C, typename IO, bool FILM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ln_bwd_kernel(const LnBwdArgs a_) {
  LnBwdArgs a = a_;
  if (a.seed_offset) { const uint64_t o = *a.seed_offset; a.seed_pre += o; a.seed_post += o; }
  constexpr int E = RowVec<C>::E, LPR = RowVec<C>::LPR, RPW = RowVec<C>::RPW;
  const IO* const DY = reinterpret_cast<const IO*>(a.dy);
  const IO* const Z = reinterpret_cast<const IO*>(a.z);
  IO* const DZ = reinterpret_cast<IO*>(a.dz);
  IO* const DA = reinterpret_cast<IO*>(a.da);
  __shared__ float red[WAVES][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l = lane % LPR, sub = lane / LPR;
  const int b = blockIdx.y;
  const int n_begin = blockIdx.x * a.rows_per_block;
  const int n_end = min(a.N, n_begin + a.rows_per_block);
  const int len_b = min(a.lens ? a.lens[b] + a.halo : a.N, a.N);
  constexpr int EF = FILM ? E : 1;                      // FiLM accumulators only where a FiLM follows the LayerNorm
  float gw[E], gb[E], gfg[EF], gfb[EF];
#pragma unroll
  for (int e = 0; e < E; ++e) gw[e] = gb[e] = 0.f;
#pragma unroll
  for (int e = 0; e < EF; ++e) gfg[e] = gfb[e] = 0.f;
  float wv[E], bv[E], fg[EF];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int c = row_col<C>(l, e);
    wv[e] = a.w[c]; bv[e] = a.bias[c];
    if constexpr (FILM) fg[e] = a.film[(size_t)b * a.ld_film + c];
  }
  for (int n0 = n_begin + wave * RPW; n0 < n_end; n0 += WAVES * RPW) {      // wave-uniform bound
    const int n = n0 + sub;
    const long row = (long)b * a.N + n;
    const bool inb = n < n_end, valid = inb && n < len_b;
    float dy[E], z[E], dzv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { dy[e] = 0.f; z[e] = 0.f; }
    float mu = 0.f, rs = 0.f;
    if (valid) {
      row_load<C>(DY + row * C, l, dy);
      row_load<C>(Z + row * C, l, z);
      mu = a.mean[row]; rs = a.rstd[row];
    }
    float s1 = 0.f, s2 = 0.f, g[E], xh[E], fpost[E];
    if (a.thresh_post) row_dropout<C>(a.seed_post, (uint64_t)row, l, a.thresh_post, a.inv_keep_post, fpost);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      xh[e] = (z[e] - mu) * rs;
      float ln = xh[e] * wv[e] + bv[e];                 // LayerNorm output (before post-dropout / FiLM)
      float d = dy[e];                                  // zero on padded / out-of-range rows: they add nothing below
      float post = 1.f;
      if (a.thresh_post) post = fpost[e];
      if constexpr (FILM) { gfg[e] += d * ln * post; gfb[e] += d; d *= fg[e]; }
      d *= post;                                        // gradient w.r.t. LN output
      gw[e] += d * xh[e]; gb[e] += d;
      g[e] = d * wv[e];
      s1 += g[e]; s2 += g[e] * xh[e];
    }
    s1 = row_sum<C>(s1) * (1.0f / C);
    s2 = row_sum<C>(s2) * (1.0f / C);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      dzv[e] = rs * (g[e] - s1 - xh[e] * s2);
      if (a.relu_mask && !(z[e] > 0.f)) dzv[e] = 0.f;
    }
    if (!inb) continue;                                 // no cross-lane traffic below this point
    row_store<C>(DZ + row * C, l, dzv);
    if ((a.da || a.dg_h) && a.thresh_pre) {
      float f[E];
      row_dropout<C>(a.seed_pre, (uint64_t)row, l, a.thresh_pre, a.inv_keep_pre, f);
#pragma unroll
      for (int e = 0; e < E; ++e) dzv[e] *= f[e];
    }
    if (a.da) row_store<C>(DA + row * C, l, dzv);
    if constexpr (sizeof(IO) == 4 && C < 256) { if (a.dg_h) row_store<C>(a.dg_h + row * C, l, dzv); }
  }
  // per-channel sums: fold the row groups of a wave, then the waves through LDS, then one atomic per channel per block
  auto reduce_and_add = [&](float* vals, float* dst) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) vals[e] += __shfl_xor(vals[e], off, 64);
    }
    __syncthreads();
    if (sub == 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) red[wave][row_col<C>(l, e)] = vals[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += WAVES * 64) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; w += 4) s += (red[w][c] + red[w + 1][c]) + (red[w + 2][c] + red[w + 3][c]);
      if (s != 0.f) atomicAdd(&dst[c], s);
    }
  };
  reduce_and_add(gw, a.dw);
  reduce_and_add(gb, a.dbias);
  if constexpr (FILM) {
    reduce_and_add(gfg, a.dfilm + (size_t)b * a.ld_dfilm);
    reduce_and_add(gfb, a.dfilm + (size_t)b * a.ld_dfilm + C);
  }
}

// ------------------------------------------------------------------------------------------------
// embedding / positional encoding / masks
// ------------------------------------------------------------------------------------------------
// out[b,n,:] = n < len ? (emb ? emb[sym[b,n]] : x[b,n,:]) + pe[n] : 0          (D = 128)
__global__ __launch_bounds__(256) void add_pos_kernel(const float* __restrict__ x, const long* __restrict__ sym, const float* __restrict__ emb,
                                                      const float* __restrict__ pe, const int* __restrict__ lens,
                                                      float* __restrict__ out, int B, int N, int D) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)B * N;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    const bool valid = n < lens[b];
    for (int c = lane * 2; c < D; c += 128) {
      float2 v = make_float2(0.f, 0.f);
      if (valid) {
        const float2 s = emb ? *reinterpret_cast<const float2*>(emb + (size_t)sym[row] * D + c)
                             : *reinterpret_cast<const float2*>(x + row * D + c);
        const float2 p = *reinterpret_cast<const float2*>(pe + (size_t)n * D + c);
        v = make_float2(s.x + p.x, s.y + p.y);
      }
      *reinterpret_cast<float2*>(out + row * D + c) = v;
    }
  }
}

// out[b,n,:] = n < len ? in[b,n,:] : 0   (masked_fill backward, pooled-gradient masks).  blockIdx.y = batch row: 32-bit index arithmetic only
// (the flat form divided a 64-bit index twice per 16-byte access)
__global__ __launch_bounds__(256) void mask_rows_kernel(const float* __restrict__ in, const int* __restrict__ lens, float* __restrict__ out,
                                                        int B, int N, int C) {
  const int b = blockIdx.y;
  const unsigned c4n = (unsigned)C >> 2, units = (unsigned)N * c4n, live = (unsigned)min(lens[b], N) * c4n;     // rows below len are the first `live` units
  const float4* src = reinterpret_cast<const float4*>(in) + (size_t)b * units;
  float4* dst = reinterpret_cast<float4*>(out) + (size_t)b * units;
  for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < units; u += gridDim.x * 256u)
    dst[u] = u < live ? src[u] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// demb[sym[b,n]] += dout[b,n,:] for valid rows
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ dout, const long* __restrict__ sym, const int* __restrict__ lens,
                                                            float* __restrict__ demb, int B, int N, int D) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)B * N;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    if (n >= lens[b]) continue;
    const long s = sym[row];
    for (int c = lane; c < D; c += 64) atomicAdd(&demb[(size_t)s * D + c], dout[row * D + c]);
  }
}

// accent-encoder input: out = mask(prenet + conv1->D(energy) + conv1->D(pitch) + pe)      D == 128
__global__ __launch_bounds__(256) void accent_sum_kernel(const float* __restrict__ prenet, const float* __restrict__ energy, const float* __restrict__ pitch,
                                                         const float* __restrict__ we, const float* __restrict__ be,
                                                         const float* __restrict__ wp, const float* __restrict__ bp,
                                                         const float* __restrict__ pe, const int* __restrict__ lens,
                                                         float* __restrict__ out, int B, int N) {
  constexpr int D = 128;
  const int lane = threadIdx.x & 63;
  const int c = lane * 2;
  float wE[2][3], wP[2][3], bE[2], bP[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int t = 0; t < 3; ++t) { wE[k][t] = we[(c + k) * 3 + t]; wP[k][t] = wp[(c + k) * 3 + t]; }
    bE[k] = be[c + k]; bP[k] = bp[c + k];
  }
  const long rows = (long)B * N;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * 4) {
    const int b = (int)(row / N), n = (int)(row - (long)b * N);
    float2 v = make_float2(0.f, 0.f);
    if (n < lens[b]) {
      const float* eb = energy + (size_t)b * N;
      const float* pb = pitch + (size_t)b * N;
      const float e0 = n > 0 ? eb[n - 1] : 0.f, e1 = eb[n], e2 = n + 1 < N ? eb[n + 1] : 0.f;
      const float p0 = n > 0 ? pb[n - 1] : 0.f, p1 = pb[n], p2 = n + 1 < N ? pb[n + 1] : 0.f;
      const float2 pr = *reinterpret_cast<const float2*>(prenet + row * D + c);
      const float2 ps = *reinterpret_cast<const float2*>(pe + (size_t)n * D + c);
      float ev[2], pv[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        ev[k] = bE[k] + (wE[k][0] * e0 + wE[k][1] * e1 + wE[k][2] * e2);
        pv[k] = bP[k] + (wP[k][0] * p0 + wP[k][1] * p1 + wP[k][2] * p2);
      }
      v = make_float2(((pr.x + ev[0]) + pv[0]) + ps.x, ((pr.y + ev[1]) + pv[1]) + ps.y);  // reference order, model.py:705
    }
    *reinterpret_cast<float2*>(out + row * D + c) = v;
  }
}

// gradients of the two Conv1d(1->D) embeddings from dout (already masked): dW[c][t] += sum dout[b,n,c] * s[b,n+t-1]
// thread = channel (128 per half block), two halves interleave rows.
__global__ __launch_bounds__(256) void scalar_conv_wgrad_kernel(const float* __restrict__ dout, int ldd, const float* __restrict__ rowscale,
                                                                const float* __restrict__ s0, const float* __restrict__ s1,
                                                                const int* __restrict__ lens,
                                                                float* __restrict__ dw0, float* __restrict__ db0,
                                                                float* __restrict__ dw1, float* __restrict__ db1,
                                                                int B, int N, int rows_per_block) {
  // thread = 4 channels (32 threads per row, 16-byte loads of dout) x one of 8 row groups; the scalar streams of the block's rows
  // (+ one halo row each side) sit in LDS.  (One channel per thread and one 4-byte load per row ran at 2 % of the HBM rate: 54 us
  // for 22 MB.)
  constexpr int D = 128;
  const int q = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int b = blockIdx.y;
  const int n_begin = blockIdx.x * rows_per_block;
  const int n_end = min(min(N, n_begin + rows_per_block), lens[b]);
  __shared__ float xs[2][256 + 2];
  __shared__ float rsc[256];
  for (int i = threadIdx.x; i < rows_per_block + 2; i += 256) {
    const int m = n_begin - 1 + i;
    const bool in = m >= 0 && m < N;
    xs[0][i] = in ? s0[(size_t)b * N + m] : 0.f;
    xs[1][i] = (in && s1) ? s1[(size_t)b * N + m] : 0.f;
  }
  for (int i = threadIdx.x; i < rows_per_block; i += 256) rsc[i] = (rowscale && n_begin + i < N) ? rowscale[(size_t)b * N + n_begin + i] : 1.f;
  __syncthreads();
  f32x4 a0[3], a1[3], ab = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 3; ++t) { a0[t] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  // The tap products are written as single v_fma_f32 instructions on purpose.  Left to the compiler, a0[1] += g * xs[0][i + 1] becomes
  // v_pk_fma_f32 ... op_sel:[0,1,0] (the LOW result reads the HIGH register of the (xs[i], xs[i + 1]) pair a ds_read2_b32 delivered), and
  // that form returned wrong low results -- the even channels >= 64 of the middle tap, a random subset per launch, 1e-2 of the sum --
  // whenever an MFMA kernel of another stream shared the CUs (round 2's "lost update"; reproduced and bisected to this instruction form
  // in round 3: tools/experiment_fork_wgrad.py).  Alone on the chip, or in this form, the kernel is exact.
#ifndef DX_SCW_PACKED
#define DX_SCW_PACKED 0
#endif
#pragma unroll 4
  for (int n = n_begin + grp; n < n_end; n += 8) {
    f32x4 g = *reinterpret_cast<const f32x4*>(dout + ((size_t)b * N + n) * ldd + q * 4);
    const int i = n - n_begin;
    g *= rsc[i];
    ab += g;
#if DX_SCW_PACKED
#pragma unroll
    for (int t = 0; t < 3; ++t) { a0[t] += g * xs[0][i + t]; a1[t] += g * xs[1][i + t]; }
#else
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const float x0 = xs[0][i + t], x1 = xs[1][i + t];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc0 = a0[t][e], acc1 = a1[t][e];
        const float ge = g[e];
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(ge), "v"(x0));
        asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(ge), "v"(x1));
        a0[t][e] = acc0; a1[t][e] = acc1;
      }
    }
#endif
  }
  // the eight row groups fold in LDS first: every atomic lands on one of ~1 k addresses and same-address atomics serialise
  __shared__ float fold[8][7][D];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    *reinterpret_cast<f32x4*>(&fold[grp][t][q * 4]) = a0[t];
    *reinterpret_cast<f32x4*>(&fold[grp][3 + t][q * 4]) = a1[t];
  }
  *reinterpret_cast<f32x4*>(&fold[grp][6][q * 4]) = ab;
  __syncthreads();
  for (int u = threadIdx.x; u < 7 * D; u += 256) {
    const int k = u / D, c = u - k * D;
    float v = 0.f;
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) v += fold[gq][k][c];
    if (v == 0.f) continue;
    if (k < 3) atomicAdd(&dw0[c * 3 + k], v);
    else if (k < 6) { if (s1) atomicAdd(&dw1[c * 3 + (k - 3)], v); }
    else { atomicAdd(&db0[c], v); if (s1) atomicAdd(&db1[c], v); }
  }
}

// pooled[b,c] = sum_n x[b,n,c] / len_b.  One block per utterance, fixed summation order (bitwise reproducible).
__global__ __launch_bounds__(1024) void mean_pool_kernel(const float* __restrict__ x, const int* __restrict__ lens, float* __restrict__ out,
                                                         int B, int N, int C) {
  __shared__ float part[1024];
  const int b = blockIdx.x;
  const int lanes = min(C, 1024);                // C == 128: eight row groups per channel (one block per utterance is all the
  const int groups = 1024 / lanes;               // parallelism a fixed summation order leaves: 16 waves instead of 4)
  const int g = threadIdx.x / lanes;
  const int rows = min(N, lens[b]);              // rows beyond the utterance are zero: adding them changes nothing
  for (int c0 = 0; c0 < C; c0 += lanes) {
    const int c = c0 + threadIdx.x % lanes;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (g < groups && c < C) {
      const float* xb = x + (size_t)b * N * C + c;
      int n = g;
      for (; n + 3 * groups < rows; n += 4 * groups) {
        s0 += xb[(size_t)n * C]; s1 += xb[(size_t)(n + groups) * C];
        s2 += xb[(size_t)(n + 2 * groups) * C]; s3 += xb[(size_t)(n + 3 * groups) * C];
      }
      for (; n < rows; n += groups) s0 += xb[(size_t)n * C];
    }
    __syncthreads();
    part[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && c < C) {
      float s = 0.f;
      for (int k = 0; k < groups; ++k) s += part[k * lanes + threadIdx.x];
      out[(size_t)b * C + c] = s / (float)lens[b];
    }
  }
}

// dx[b,n,c] = n < len ? dout[b,c] / len_b : 0.  blockIdx.y = batch row, 16 bytes per access, 32-bit index arithmetic (the flat scalar form
// took a 64-bit modulo and two 64-bit divisions per ELEMENT: 13.5 us for 22 MB)
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ lens, float* __restrict__ dx,
                                                            int B, int N, int C) {
  const int b = blockIdx.y;
  const unsigned c4n = (unsigned)C >> 2, units = (unsigned)N * c4n;
  const int len = lens[b];
  const unsigned live = (unsigned)min(len, N) * c4n;
  const float flen = (float)len;
  float4* dst = reinterpret_cast<float4*>(dx) + (size_t)b * units;
  for (unsigned u = blockIdx.x * 256u + threadIdx.x; u < units; u += gridDim.x * 256u) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u < live) {
      const float4 g = *reinterpret_cast<const float4*>(dout + (size_t)b * C + (u % c4n) * 4);
      v = make_float4(g.x / flen, g.y / flen, g.z / flen, g.w / flen);
    }
    dst[u] = v;
  }
}

// batched transpose in[b][R][Cc] -> out[b][Cc][R]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc, int accumulate) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* ib = in + (size_t)b * R * Cc;
  float* ob = out + (size_t)b * R * Cc;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < Cc) ? ib[(size_t)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < Cc && r < R) ob[(size_t)c * R + r] = accumulate ? ob[(size_t)c * R + r] + tile[tx][k] : tile[tx][k];
  }
}

// y = x / max(||x||_2, 1e-12) per row (F.normalize)
__global__ __launch_bounds__(64) void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ y, int C) {
  const int row = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[(size_t)row * C + c]; s += v * v; }
  const float nrm = fmaxf(sqrtf(dx_wave_sum(s)), 1e-12f);
  for (int c = lane; c < C; c += 64) y[(size_t)row * C + c] = x[(size_t)row * C + c] / nrm;
}

// mean cross entropy over B rows of S logits; dlogits = (softmax - onehot) / B.  One block.
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, const long* __restrict__ target,
                                                            float* __restrict__ loss, float* __restrict__ dlogits, int B, int S) {
  __shared__ float part[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* l = logits + (size_t)b * S;
    float m = -INFINITY;
    for (int s = 0; s < S; ++s) m = fmaxf(m, l[s]);
    float z = 0.f;
    for (int s = 0; s < S; ++s) z += expf(l[s] - m);
    const float lse = m + logf(z);
    acc += lse - l[target[b]];
    for (int s = 0; s < S; ++s) dlogits[(size_t)b * S + s] = (expf(l[s] - lse) - (s == target[b] ? 1.f : 0.f)) / (float)B;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = part[0] / (float)B;
}

// out = dy * (y > 0)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// out[row][c] = x[row][c] * scale[c] + shift[c]     (eval-mode BatchNorm of the frozen pitch predictor); IO = float or the 16-bit type
template <typename IO>
__global__ __launch_bounds__(256) void channel_affine_kernel(const IO* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                             IO* __restrict__ out, unsigned total_units, unsigned units_per_row) {
  constexpr int V = sizeof(IO) == 4 ? 4 : 8;          // elements per 16-byte access
  // the host picks a grid whose stride (gridDim.x * 256 units) is a multiple of a row: a thread stays on ONE channel group and keeps its
  // scale / shift in registers (a 64-bit `% C` per element had made the 16-bit form slower than the fp32 one: 40 vs 17 us)
  const unsigned first = blockIdx.x * 256u + threadIdx.x;
  const int c = (int)(first % units_per_row) * V;
  float sc[V], sh[V];
#pragma unroll
  for (int e = 0; e < V; e += 4) {
    const float4 a = *reinterpret_cast<const float4*>(scale + c + e), b = *reinterpret_cast<const float4*>(shift + c + e);
    sc[e] = a.x; sc[e + 1] = a.y; sc[e + 2] = a.z; sc[e + 3] = a.w;
    sh[e] = b.x; sh[e + 1] = b.y; sh[e + 2] = b.z; sh[e + 3] = b.w;
  }
  for (unsigned i = first; i < total_units; i += gridDim.x * 256u) {
    if constexpr (sizeof(IO) == 4) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      reinterpret_cast<float4*>(out)[i] = make_float4(v.x * sc[0] + sh[0], v.y * sc[1] + sh[1], v.z * sc[2] + sh[2], v.w * sc[3] + sh[3]);
    } else {
      const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (dx_h16)((float)v[e] * sc[e] + sh[e]);
      reinterpret_cast<bf16x8*>(out)[i] = o;
    }
  }
}

// FiLM parameters of all FFT blocks from the two predictor outputs (StyleAdapter.forward, model.py:779-800):
//   film[blk][b][0:C] = pm[0][blk] * gammas[b][blk*C + c] + 1,   film[blk][b][C:2C] = pm[1][blk] * betas[b][blk*C + c]      (pm = null: 1)
// written block-major, (B, 2C) contiguous per block: what the LayerNorm kernels read.  One launch instead of mul, add, mul, cat, transpose.
__global__ __launch_bounds__(256) void film_affine_fwd_kernel(const float* __restrict__ gammas, const float* __restrict__ betas, const float* __restrict__ pm,
                                                              float* __restrict__ film, int B, int nb, int C) {
  const long total = (long)nb * B * 2 * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c2 = (int)(i % (2 * C));
    const int b = (int)((i / (2 * C)) % B);
    const int blk = (int)(i / ((long)2 * C * B));
    const bool beta = c2 >= C;
    const int c = beta ? c2 - C : c2;
    const float m = pm ? pm[(beta ? nb : 0) + blk] : 1.f;
    const float v = (beta ? betas : gammas)[(size_t)b * nb * C + blk * C + c];
    film[i] = beta ? m * v : m * v + 1.f;
  }
}

// Backward: dgammas = pm[0][blk] * dfilm[blk][:, 0:C], dbetas likewise, dpm[0][blk] += sum dfilm_gamma * gammas, dpm[1][blk] += sum dfilm_beta * betas.
// One block of 256 threads per (FFT block, half): the 2 nb scalar sums fold in LDS, one atomic each.  A null dfilm pointer = zero gradient.
struct FilmBwdArgs {
  const float* dfilm[16];        // per FFT block: (B, 2C) contiguous, or null
  const float* gammas; const float* betas; const float* pm;
  float* dgammas; float* dbetas; float* dpm;
  int B, nb, C;
};
__global__ __launch_bounds__(256) void film_affine_bwd_kernel(const FilmBwdArgs a) {
  // block = (FFT block, gamma | beta, 8 utterances): 4 independent elements per thread (one block per (block, half) walked B x C elements
  // with a division and a dependent load chain per step: 24 us for 100 KB)
  const int blk = blockIdx.x, beta = blockIdx.y, b0 = blockIdx.z * 8;
  const float* df = a.dfilm[blk];
  const float* src = beta ? a.betas : a.gammas;
  float* dst = beta ? a.dbetas : a.dgammas;
  const float m = a.pm ? a.pm[(beta ? a.nb : 0) + blk] : 1.f;
  float acc = 0.f;
  const int nb_rows = min(8, a.B - b0);
  for (int i = threadIdx.x; i < nb_rows * a.C; i += 256) {
    const int b = b0 + i / a.C, c = i % a.C;
    const size_t o = (size_t)b * a.nb * a.C + blk * a.C + c;
    const float d = df ? df[(size_t)b * 2 * a.C + (beta ? a.C : 0) + c] : 0.f;
    dst[o] = m * d;
    acc += d * src[o];
  }
  __shared__ float part[4];
  acc = dx_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && a.dpm) {
    const float t = (part[0] + part[1]) + (part[2] + part[3]);
    if (t != 0.f) atomicAdd(&a.dpm[(beta ? a.nb : 0) + blk], t);
  }
}

// per-speaker z-normalisation that preserves exact zeros (silence / unvoiced): dynamic_stats.py:156-183
// table[s] = {energy mean, energy std, pitch mean, pitch std}; valid[s] = 0 -> utterance left untouched
__global__ __launch_bounds__(256) void condition_prosody_kernel(const float* __restrict__ in, float* __restrict__ out, const long* __restrict__ spk,
                                                                const float* __restrict__ table, const int* __restrict__ valid,
                                                                int which, int N, int S) {
  const int b = blockIdx.y;
  const long s = spk[b];
  const bool ok = s >= 0 && s < S && valid[s];
  const float mean = ok ? table[s * 4 + which * 2] : 0.f, sd = ok ? table[s * 4 + which * 2 + 1] : 1.f;
  for (int n = blockIdx.x * 256 + threadIdx.x; n < N; n += gridDim.x * 256) {
    const float v = in[(size_t)b * N + n];
    out[(size_t)b * N + n] = ok ? (v == 0.f ? 0.f : (v - mean) / sd) : v;
  }
}

// out[b,:] = valid[spk[b]] ? emb[spk[b],:] : 0     (support-set mean speaker embedding, dynamic_stats.py:185-186)
__global__ __launch_bounds__(256) void gather_speaker_rows_kernel(const float* __restrict__ emb, const long* __restrict__ spk, const int* __restrict__ valid,
                                                                  float* __restrict__ out, int E, int S) {
  const int b = blockIdx.x;
  const long s = spk[b];
  const bool ok = s >= 0 && s < S && valid[s];
  for (int e = threadIdx.x; e < E; e += 256) out[(size_t)b * E + e] = ok ? emb[(size_t)s * E + e] : 0.f;
}

inline int row_grid(long rows) { return (int)std::min<long>((rows + 3) / 4, 8192); }

}  // namespace

extern "C" {

int dx_ln_fwd(void* av, const void* resv, const float* w, const float* bias, const float* film, int ld_film,
              const int* lens, int halo, void* yv, float* mean, float* rstd, int B, int N, int C,
              uint64_t seed_pre, float p_pre, uint64_t seed_post, float p_post, const uint64_t* seed_offset, int io_bf16, void* y_bf16_copy,
              void* stream) {
  float* a = (float*)av; const float* res = (const float*)resv; float* y = (float*)yv;
  DX_REQUIRE(!y_bf16_copy || (C == 128 && !io_bf16), "dx_ln_fwd: the bf16 shadow output is for fp32 rows with C = 128");
  DX_REQUIRE(a && w && bias && y && mean && rstd, "dx_ln_fwd: null pointer");
  DX_REQUIRE(!io_bf16 || C == 1024, "dx_ln_fwd: bf16 rows are supported for C = 1024 only");
  DX_REQUIRE(C == 128 || C == 1024, "dx_ln_fwd: C must be 128 or 1024 (got %d)", C);
  DX_REQUIRE(B > 0 && N > 0, "dx_ln_fwd: bad dims");
  DX_REQUIRE(p_pre >= 0.f && p_pre < 1.f && p_post >= 0.f && p_post < 1.f, "dx_ln_fwd: dropout p out of range");
  DX_REQUIRE(!film || ld_film >= 2 * C, "dx_ln_fwd: ld_film too small");
  LnArgs k{a, res, w, bias, film, ld_film, lens, halo, y, (dx_h16*)y_bf16_copy, mean, rstd, B, N,
           seed_pre, (uint32_t)lrintf(p_pre * 65536.f), 1.f / (1.f - p_pre),
           seed_post, (uint32_t)lrintf(p_post * 65536.f), 1.f / (1.f - p_post), seed_offset};
  hipStream_t s = (hipStream_t)stream;
  dx_prof_begin(DX_PROF_ROWS, s);
  const int grid = row_grid((long)B * N);
  if (C == 128) hipLaunchKernelGGL((ln_fwd_kernel<128, float>), dim3(dx_cdiv(grid, RowVec<128>::RPW)), dim3(256), 0, s, k);
  else if (io_bf16) hipLaunchKernelGGL((ln_fwd_kernel<1024, dx_h16>), dim3(grid), dim3(256), 0, s, k);
  else hipLaunchKernelGGL((ln_fwd_kernel<1024, float>), dim3(grid), dim3(256), 0, s, k);
  dx_prof_end(DX_PROF_ROWS, s);
  DX_LAUNCH_CHECK("dx_ln_fwd");
  return DX_OK;
}

// z = dropout(X W^T + proj_bias) + res, y = mask(FiLM(LayerNorm(z))): the out-projection and the LayerNorm of an FFT block in one launch.
// X: 16-bit [B*N][ldx] (128 columns), Wpack: forward pack of the (128, 128) weight; every other argument as in dx_ln_fwd (C = 128, fp32 rows).
int dx_proj_ln_fwd(const void* X, int ldx, const void* Wpack, const float* proj_bias, float* z, const float* res, const float* w, const float* bias,
                   const float* film, int ld_film, const int* lens, int halo, float* y, float* mean, float* rstd, int B, int N,
                   uint64_t seed_pre, float p_pre, const uint64_t* seed_offset, void* y_bf16_copy, void* stream) {
  DX_REQUIRE(X && Wpack && z && w && bias && y && mean && rstd, "dx_proj_ln_fwd: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && ldx >= 128 && (ldx % 8) == 0, "dx_proj_ln_fwd: bad dims");
  DX_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)Wpack % 16) == 0 && ((uintptr_t)z % 16) == 0 && ((uintptr_t)y % 16) == 0, "dx_proj_ln_fwd: pointers must be 16-byte aligned");
  DX_REQUIRE(p_pre >= 0.f && p_pre < 1.f, "dx_proj_ln_fwd: dropout p out of range");
  DX_REQUIRE(!film || ld_film >= 256, "dx_proj_ln_fwd: ld_film too small");
  ProjLnArgs k{(const dx_h16*)X, ldx, (const dx_h16*)Wpack, proj_bias,
               LnArgs{z, res, w, bias, film, ld_film, lens, halo, y, (dx_h16*)y_bf16_copy, mean, rstd, B, N,
                      seed_pre, (uint32_t)lrintf(p_pre * 65536.f), 1.f / (1.f - p_pre), 0, 0u, 1.f, seed_offset}};
  hipStream_t s = (hipStream_t)stream;
  dx_prof_begin(DX_PROF_ROWS, s);
  hipLaunchKernelGGL(proj_ln_fwd_kernel, dim3((unsigned)(((long)B * N + 63) / 64)), dim3(256), 0, s, k);
  dx_prof_end(DX_PROF_ROWS, s);
  DX_LAUNCH_CHECK("dx_proj_ln_fwd");
  return DX_OK;
}

int dx_ln_bwd(const void* dyv, const void* zv, const float* mean, const float* rstd, const float* w, const float* bias,
              const float* film, int ld_film, const int* lens, int halo, void* dzv, void* dav, float* dw, float* dbias,
              float* dfilm, int ld_dfilm, int B, int N, int C, int relu_mask,
              uint64_t seed_pre, float p_pre, uint64_t seed_post, float p_post, const uint64_t* seed_offset, int io_bf16, void* dg_bf16_copy,
              void* stream) {
  const float* dy = (const float*)dyv;
  DX_REQUIRE(!dg_bf16_copy || (C == 128 && !io_bf16), "dx_ln_bwd: the bf16 shadow output is for fp32 rows with C = 128"); const float* z = (const float*)zv; float* dz = (float*)dzv; float* da = (float*)dav;
  DX_REQUIRE(!io_bf16 || C == 1024, "dx_ln_bwd: bf16 rows are supported for C = 1024 only");
  DX_REQUIRE(dy && z && mean && rstd && w && bias && dz && dw && dbias, "dx_ln_bwd: null pointer");
  DX_REQUIRE(C == 128 || C == 1024, "dx_ln_bwd: C must be 128 or 1024 (got %d)", C);
  DX_REQUIRE((film == nullptr) == (dfilm == nullptr), "dx_ln_bwd: film and dfilm must come together");
  // every block ends in one atomic per channel on the SAME 2 C addresses (dw, dbias): same-address atomics serialise (~15 ns each), so the
  // block count is a trade between memory parallelism in the row loop and the length of that atomic chain
  static const int rpb_env = getenv("DX_LN_BWD_RPB") ? atoi(getenv("DX_LN_BWD_RPB")) : 0;
  static const int big_env = getenv("DX_LN_BWD_BIG") ? atoi(getenv("DX_LN_BWD_BIG")) : 1;
  // sixteen-wave blocks for C = 128 once there are rows enough to fill the chip with them (C = 1024 would need 64 KB of LDS per block: slower).
  // Measured at C2 (frame axis, 43 k rows): C = 128: 4 waves x 64 rows 25.4 us, 16 waves x 256 rows 23.6, 16 x 192 21.4; C = 1024: 4 x 64 94 us, 4 x 96 78
  const bool big = big_env && C == 128 && (long)B * N >= 16384;
  static const int rpb_1024 = getenv("DX_LN_BWD_RPB_1024") ? atoi(getenv("DX_LN_BWD_RPB_1024")) : 32;   // 96 / 48 / 32 / 24 rows: 6.363 / 6.350 / 6.345 / 6.351 ms per step
  static const int rpb_small = getenv("DX_LN_BWD_RPB_SMALL") ? atoi(getenv("DX_LN_BWD_RPB_SMALL")) : 64;
  const int rpb = rpb_env > 0 ? rpb_env : (big ? 192 : (C == 1024 ? rpb_1024 : rpb_small));
  LnBwdArgs k{dy, z, mean, rstd, w, bias, film, ld_film, lens, halo, dz, da, (dx_h16*)dg_bf16_copy, dw, dbias, dfilm, ld_dfilm, B, N, rpb, relu_mask,
              seed_pre, (uint32_t)lrintf(p_pre * 65536.f), 1.f / (1.f - p_pre),
              seed_post, (uint32_t)lrintf(p_post * 65536.f), 1.f / (1.f - p_post), seed_offset};
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(dx_cdiv(N, rpb), B);
  dx_prof_begin(DX_PROF_ROWS, s);
#define DX_LN_BWD(CC, IO) do { \
    if (big) { \
      if (film) hipLaunchKernelGGL((ln_bwd_kernel<CC, IO, true, 16>), grid, dim3(1024), 0, s, k); \
      else hipLaunchKernelGGL((ln_bwd_kernel<CC, IO, false, 16>), grid, dim3(1024), 0, s, k); \
    } else { \
      if (film) hipLaunchKernelGGL((ln_bwd_kernel<CC, IO, true, 4>), grid, dim3(256), 0, s, k); \
      else hipLaunchKernelGGL((ln_bwd_kernel<CC, IO, false, 4>), grid, dim3(256), 0, s, k); } } while (0)
  if (C == 128) DX_LN_BWD(128, float);
  else if (io_bf16) DX_LN_BWD(1024, dx_h16);
  else DX_LN_BWD(1024, float);
#undef DX_LN_BWD
  dx_prof_end(DX_PROF_ROWS, s);
  DX_LAUNCH_CHECK("dx_ln_bwd");
  return DX_OK;
}

// out = mask(emb[sym] + pe)  (emb != null)   or   out = mask(x + pe)  (emb == null)
int dx_add_pos(const float* x, const long* sym, const float* emb, const float* pe, const int* lens, float* out,
               int B, int N, int D, int pe_rows, void* stream) {
  DX_REQUIRE(pe && lens && out && (emb ? (sym != nullptr) : (x != nullptr)), "dx_add_pos: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && D > 0 && (D % 2) == 0, "dx_add_pos: bad dims");
  DX_REQUIRE(N <= pe_rows, "dx_add_pos: sequence length %d exceeds the positional table (%d rows)", N, pe_rows);
  hipLaunchKernelGGL(add_pos_kernel, dim3(row_grid((long)B * N)), dim3(256), 0, (hipStream_t)stream, x, sym, emb, pe, lens, out, B, N, D);
  DX_LAUNCH_CHECK("dx_add_pos");
  return DX_OK;
}

int dx_mask_rows(const float* in, const int* lens, float* out, int B, int N, int C, void* stream) {
  DX_REQUIRE(in && lens && out && B > 0 && N > 0 && C > 0 && (C % 4) == 0, "dx_mask_rows: bad arguments");
  DX_REQUIRE((long)N * C / 4 < (1l << 31), "dx_mask_rows: a batch row must have fewer than 2^31 16-byte units");
  const long units = (long)N * C / 4;
  hipLaunchKernelGGL(mask_rows_kernel, dim3((int)std::min<long>((units + 255) / 256, std::max<long>(1, 8192 / B)), B), dim3(256), 0, (hipStream_t)stream, in, lens, out, B, N, C);
  DX_LAUNCH_CHECK("dx_mask_rows");
  return DX_OK;
}

int dx_embedding_bwd(const float* dout, const long* sym, const int* lens, float* demb, int B, int N, int D, void* stream) {
  DX_REQUIRE(dout && sym && lens && demb && B > 0 && N > 0 && D > 0, "dx_embedding_bwd: bad arguments");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(row_grid((long)B * N)), dim3(256), 0, (hipStream_t)stream, dout, sym, lens, demb, B, N, D);
  DX_LAUNCH_CHECK("dx_embedding_bwd");
  return DX_OK;
}

int dx_accent_sum(const float* prenet, const float* energy, const float* pitch, const float* we, const float* be,
                  const float* wp, const float* bp, const float* pe, const int* lens, float* out,
                  int B, int N, int D, int pe_rows, void* stream) {
  DX_REQUIRE(prenet && energy && pitch && we && be && wp && bp && pe && lens && out, "dx_accent_sum: null pointer");
  DX_REQUIRE(D == 128, "dx_accent_sum: hidden dim must be 128 (got %d)", D);
  DX_REQUIRE(N <= pe_rows, "dx_accent_sum: sequence length %d exceeds the positional table (%d rows)", N, pe_rows);
  hipLaunchKernelGGL(accent_sum_kernel, dim3(row_grid((long)B * N)), dim3(256), 0, (hipStream_t)stream,
                     prenet, energy, pitch, we, be, wp, bp, pe, lens, out, B, N);
  DX_LAUNCH_CHECK("dx_accent_sum");
  return DX_OK;
}

// dW0[c][t] += sum_{b, n < len_b} rowscale[b,n] * dout[b,n,c] * s0[b,n+t-1]; db0[c] += ...; same for (s1, dW1, db1) if s1.
int dx_scalar_conv_wgrad(const float* dout, int ldd, const float* rowscale, const float* s0, const float* s1, const int* lens,
                         float* dw0, float* db0, float* dw1, float* db1, int B, int N, int D, void* stream) {
  DX_REQUIRE(dout && s0 && lens && dw0 && db0, "dx_scalar_conv_wgrad: null pointer");
  DX_REQUIRE(D == 128 && ldd >= 0, "dx_scalar_conv_wgrad: hidden dim must be 128");  // ldd == 0: one row broadcast
  DX_REQUIRE(!s1 || (dw1 && db1), "dx_scalar_conv_wgrad: second stream needs its gradient buffers");
  static const int rpb_env = getenv("DX_SCW_RPB") ? atoi(getenv("DX_SCW_RPB")) : 0;
  // every block ends in 7 x 128 atomics on the same ~900 addresses, and same-address atomics cost ~70 ns each when all blocks finish together:
  // the launch time is proportional to the number of blocks (32 / 64 / 128 / 256 rows per block: 95 / 51 / 30 / 20 us on the frame axis)
  const int rpb = rpb_env > 0 ? std::min(rpb_env, 256) : 256;
  hipLaunchKernelGGL(scalar_conv_wgrad_kernel, dim3(dx_cdiv(N, rpb), B), dim3(256), 0, (hipStream_t)stream,
                     dout, ldd, rowscale, s0, s1, lens, dw0, db0, dw1, db1, B, N, rpb);
  DX_LAUNCH_CHECK("dx_scalar_conv_wgrad");
  return DX_OK;
}

int dx_mean_pool(const float* x, const int* lens, float* out, int B, int N, int C, void* stream) {
  DX_REQUIRE(x && lens && out && B > 0 && N > 0 && C > 0, "dx_mean_pool: bad arguments");
  hipLaunchKernelGGL(mean_pool_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, x, lens, out, B, N, C);
  DX_LAUNCH_CHECK("dx_mean_pool");
  return DX_OK;
}

int dx_mean_pool_bwd(const float* dout, const int* lens, float* dx, int B, int N, int C, void* stream) {
  DX_REQUIRE(dout && lens && dx && B > 0 && N > 0 && C > 0 && (C % 4) == 0 && (long)N * C / 4 < (1l << 31), "dx_mean_pool_bwd: bad arguments (C must be a multiple of 4)");
  const long units = (long)N * C / 4;
  hipLaunchKernelGGL(mean_pool_bwd_kernel, dim3((int)std::min<long>((units + 255) / 256, std::max<long>(1, 8192 / B)), B), dim3(256), 0, (hipStream_t)stream, dout, lens, dx, B, N, C);
  DX_LAUNCH_CHECK("dx_mean_pool_bwd");
  return DX_OK;
}

int dx_transpose(const float* in, float* out, int B, int R, int Cc, int accumulate, void* stream) {
  DX_REQUIRE(in && out && B > 0 && R > 0 && Cc > 0, "dx_transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3(dx_cdiv(Cc, 32), dx_cdiv(R, 32), B), dim3(256), 0, (hipStream_t)stream, in, out, R, Cc, accumulate);
  DX_LAUNCH_CHECK("dx_transpose");
  return DX_OK;
}

int dx_l2_normalize(const float* x, float* y, int rows, int C, void* stream) {
  DX_REQUIRE(x && y && rows > 0 && C > 0, "dx_l2_normalize: bad arguments");
  hipLaunchKernelGGL(l2_normalize_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, x, y, C);
  DX_LAUNCH_CHECK("dx_l2_normalize");
  return DX_OK;
}

int dx_cross_entropy(const float* logits, const long* target, float* loss, float* dlogits, int B, int S, void* stream) {
  DX_REQUIRE(logits && target && loss && dlogits && B > 0 && S > 0, "dx_cross_entropy: bad arguments");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, loss, dlogits, B, S);
  DX_LAUNCH_CHECK("dx_cross_entropy");
  return DX_OK;
}

int dx_relu_bwd(const float* dy, const float* y, float* out, long n, void* stream) {
  DX_REQUIRE(dy && y && out && n > 0, "dx_relu_bwd: bad arguments");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((int)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, (hipStream_t)stream, dy, y, out, n);
  DX_LAUNCH_CHECK("dx_relu_bwd");
  return DX_OK;
}

int dx_channel_affine(const void* x, const float* scale, const float* shift, void* out, long rows, int C, int io_bf16, void* stream) {
  const int V = io_bf16 ? 8 : 4;
  DX_REQUIRE(x && scale && shift && out && rows > 0 && C > 0 && (C % V) == 0 && rows * (C / V) < (1l << 31),
             "dx_channel_affine: bad arguments (C must be a multiple of the 16-byte access; at most 2^31 accesses)");
  const unsigned units_per_row = C / V, total = (unsigned)(rows * units_per_row);
  // grid stride = a whole number of rows: blocks = k * lcm(256, units_per_row) / 256
  unsigned a = 256, b = units_per_row;
  while (b) { const unsigned t = a % b; a = b; b = t; }                 // a = gcd(256, units_per_row)
  const unsigned step = units_per_row / a;                               // blocks per whole-row stride
  unsigned blocks = std::min<unsigned>((total + 255) / 256, 4096);
  blocks = std::max(step, blocks / step * step);
  if (io_bf16) hipLaunchKernelGGL(channel_affine_kernel<dx_h16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const dx_h16*)x, scale, shift, (dx_h16*)out, total, units_per_row);
  else hipLaunchKernelGGL(channel_affine_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, scale, shift, (float*)out, total, units_per_row);
  DX_LAUNCH_CHECK("dx_channel_affine");
  return DX_OK;
}

// film (nb, B, 2C) block-major from gammas / betas (B, nb*C) and the scalar post-multipliers pm (2, nb) or null: model.py:779-800
int dx_film_affine_fwd(const float* gammas, const float* betas, const float* pm, float* film, int B, int nb, int C, void* stream) {
  DX_REQUIRE(gammas && betas && film && B > 0 && nb > 0 && C > 0, "dx_film_affine_fwd: bad arguments");
  const long total = (long)nb * B * 2 * C;
  hipLaunchKernelGGL(film_affine_fwd_kernel, dim3((int)std::min<long>((total + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream, gammas, betas, pm, film, B, nb, C);
  DX_LAUNCH_CHECK("dx_film_affine_fwd");
  return DX_OK;
}

// dfilm_ptrs: HOST array of nb device pointers ((B, 2C) contiguous each; null = that block received no gradient).  dpm (2, nb) accumulates.
int dx_film_affine_bwd(const void* dfilm_ptrs, const float* gammas, const float* betas, const float* pm, float* dgammas, float* dbetas, float* dpm,
                       int B, int nb, int C, void* stream) {
  DX_REQUIRE(dfilm_ptrs && gammas && betas && dgammas && dbetas && B > 0 && nb > 0 && nb <= 16 && C > 0, "dx_film_affine_bwd: bad arguments (at most 16 blocks)");
  DX_REQUIRE((pm == nullptr) == (dpm == nullptr), "dx_film_affine_bwd: pm and dpm must come together");
  FilmBwdArgs a{};
  for (int i = 0; i < nb; ++i) a.dfilm[i] = reinterpret_cast<const float* const*>(dfilm_ptrs)[i];
  a.gammas = gammas; a.betas = betas; a.pm = pm; a.dgammas = dgammas; a.dbetas = dbetas; a.dpm = dpm; a.B = B; a.nb = nb; a.C = C;
  hipLaunchKernelGGL(film_affine_bwd_kernel, dim3(nb, 2, dx_cdiv(B, 8)), dim3(256), 0, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK("dx_film_affine_bwd");
  return DX_OK;
}

int dx_condition_prosody(const float* in, float* out, const long* speaker_ids, const float* table, const int* valid,
                         int which, int B, int N, int S, void* stream) {
  DX_REQUIRE(in && out && speaker_ids && table && valid && (which == 0 || which == 1) && B > 0 && N > 0 && S > 0, "dx_condition_prosody: bad arguments");
  hipLaunchKernelGGL(condition_prosody_kernel, dim3(std::max(1, std::min(dx_cdiv(N, 256), 16)), B), dim3(256), 0, (hipStream_t)stream,
                     in, out, speaker_ids, table, valid, which, N, S);
  DX_LAUNCH_CHECK("dx_condition_prosody");
  return DX_OK;
}

int dx_gather_speaker_rows(const float* emb, const long* speaker_ids, const int* valid, float* out, int B, int E, int S, void* stream) {
  DX_REQUIRE(emb && speaker_ids && valid && out && B > 0 && E > 0 && S > 0, "dx_gather_speaker_rows: bad arguments");
  hipLaunchKernelGGL(gather_speaker_rows_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, emb, speaker_ids, valid, out, E, S);
  DX_LAUNCH_CHECK("dx_gather_speaker_rows");
  return DX_OK;
}

}  // extern "C"

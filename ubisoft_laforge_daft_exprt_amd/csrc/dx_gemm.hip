// Conv1D(k=1|3, stride 1, zero 'same' padding) on channels-last activations as an MFMA GEMM, gfx950.
//
// Replaces the reference's ConvNorm1D / LinearNorm / in_proj / out_proj call sites
// (src/daft_exprt/model.py:57-94, :165-186, :206-217, :649-671) and their autograd backward.
//
//   forward      Y[b,n,co]  = epi( bias[co] + sum_tap sum_ci X[b, n+tap-PAD, ci] * W[co,ci,tap] )
//   input grad   the same kernel fed with the flipped/transposed pack of W
//   weight grad  G[tap][co][ci] += sum_{b,n} dY[b,n,co] * X[b, n+tap-PAD, ci]      (split over tokens, fp32 atomics)
//
// Data layout in HBM: activations are fp32 [B*N][C] (channels contiguous).  Weights are re-packed once per
// optimiser step from the checkpoint layout (Cout, Cin, TAPS) into [TAPS][CoutP][CinP] with Cin contiguous
// (CoutP, CinP = dims rounded up to the tile, zero filled), as f32 (exact-f32 MFMA, v_mfma_f32_16x16x4_f32)
// or bf16 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Tiling (wave64, 4 waves = 2x2 per workgroup): a workgroup owns 128 output channels x 128 tokens of ONE
// batch row; a wave owns 64x64 as 4x4 MFMA tiles of 16x16.  W is the MFMA "A" operand (rows = channels),
// X the "B" operand (columns = tokens), so every lane ends up with 4 consecutive output channels of one
// token: 16-byte stores, and per-channel epilogue vectors are 16-byte loads.  Both operands sit in LDS as
// [row][128 bytes of K] (XOR-swizzled 16-byte slots); a lane fetches its whole fragment for 16 (f32) / 32 (bf16) K values
// with one ds_read_b128.  For f32 the K index inside a 16-wide group is permuted (lane group g, element j
// <-> k = 4g + j) identically for both operands, which leaves the sum unchanged.
// The token tile is staged once per K chunk WITH a one-row halo on both sides; the three taps read it at
// row offsets 0/1/2, so X is fetched once, not three times.  Rows outside [0, N) of the batch row are
// zero, which is exactly the reference's zero 'same' padding on the padded (B, N_max) grid.
#include "dx_common.h"
#include <algorithm>
#include <stdlib.h>

namespace {

constexpr int TILE = 128;   // channels and tokens per workgroup
constexpr int DK_MAX_B = 1024;   // batch rows the persistent / deep-K kernels can number live token tiles for (larger batches: plain order)

struct ConvGemmArgs {
  const float* X; int ldx;
  const void* Wp;
  const float* bias;
  float* Y; int ldy;
  int B, N, Cin, Cout, CinP, CoutP;
  int relu;
  const float* post_scale; const float* post_shift;
  const float* relu_aux; int ld_aux;
  int accumulate;
  const int* lens; int mask_rows;
  float out_scale;
  int skip_halo;   // >= 0: token tiles starting at or beyond min(len_b + skip_halo, N) are not computed (zero-filled)
  int x_bf16, y_bf16, aux_bf16;   // storage type of X / Y / relu_aux (bf16 operand mode only); ld* are in elements
  // Optional [B]: input rows n >= rows_exist[b] of batch row b DO NOT EXIST (they read as the conv's zero padding), although the
  // buffers have N rows per batch row.  NULL = every row has N rows (the reference's padded grid).  Lets one allocation (and one
  // captured graph) serve batches whose logical padded length is smaller than N, and lets a batch row behave as if it were alone.
  const int* rows_exist;
  int xcd_swizzle;  // conv_dk, several output-channel blocks per token tile: workgroups are renumbered so that a token tile's blocks run together on one XCD
};

template <typename T> struct Mma;
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const float4& a, const float4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
  }
};
template <> struct Mma<dx_h16> {
  static __device__ __forceinline__ void run(const float4& a, const float4& b, f32x4& c) {
    c = DX_MFMA_H16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
  }
};

__device__ __forceinline__ uint2 pack_bf16x4v(const f32x4& v) {
  bf16x4 h;
  h[0] = (dx_h16)v[0]; h[1] = (dx_h16)v[1]; h[2] = (dx_h16)v[2]; h[3] = (dx_h16)v[3];
  return __builtin_bit_cast(uint2, h);
}

__device__ __forceinline__ uint2 pack_bf16x4(const float4& v) {
  bf16x4 h;
  h[0] = (dx_h16)v.x; h[1] = (dx_h16)v.y; h[2] = (dx_h16)v.z; h[3] = (dx_h16)v.w;
  return __builtin_bit_cast(uint2, h);
}

// LDS image of both operands: unpadded 128-byte rows, 16-byte slot index XOR-ed with (row & 7).  The fragment read of a
// 16-lane ds_read_b128 group then touches 16 distinct bank slots for every tap offset (0/1/2 rows): conflict free, and the
// tile needs 65 KB instead of 74 KB (two workgroups per CU with room to spare).
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

// bf16 weight packs are fragment-major: the [rows][K] matrix of a tap is cut into 16-row x 32-k blocks, each stored as the
// 64 x 8 elements one v_mfma_f32_16x16x32_bf16 "A" operand takes (lane = (k / 8 % 4) * 16 + row % 16, 8 consecutive k per lane).
// A wave reads a whole fragment with one contiguous 1 KB load; kernels that stage weights through LDS fetch the same 16-byte
// units (wb_off is a multiple of 8 for k % 8 == 0).  rowsP % 16 == 0, kP % 32 == 0.
__device__ __forceinline__ size_t wb_off(int tap, int row, int k, int rowsP, int kP) {
  return ((size_t)(tap * (rowsP >> 4) + (row >> 4)) * (kP >> 5) + (k >> 5)) * 512 + ((((k >> 3) & 3) << 4) + (row & 15)) * 8 + (k & 7);
}
// staging slot u of a [tap][128 rows][64 k] weight tile -> (tile row, 16-byte slot): consecutive lanes walk one block of the
// fragment-major pack (bf16, contiguous 1 KB per wave-instruction) or one row (f32)
template <typename T> __device__ __forceinline__ void w_unit(int u, int& row, int& q) {
  if constexpr (sizeof(T) == 2) {
    const int v = u & 1023;
    row = (u >> 10) * TILE + (v >> 7) * 16 + (v & 15);
    q = ((v >> 6) & 1) * 4 + ((v >> 4) & 3);
  } else {
    row = u >> 3; q = u & 7;
  }
}

// TOK = tokens per workgroup (128, or 64 for narrow outputs that would otherwise launch fewer workgroups than CUs).
// XH  = the activation tensor is stored as bf16 (bf16 operand mode only).
// Software pipeline: the global loads of K-chunk c+1 are issued into registers before the MFMAs of chunk c and written
// to LDS after them, so HBM/L2 latency hides under the matrix work of the same wave.
template <typename T, int TAPS, int TOK, bool XH>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvGemmArgs a) {
  constexpr int PAD = (TAPS - 1) / 2;
  constexpr int BK = 128 / (int)sizeof(T);
  constexpr int XROWS = TOK + TAPS - 1;
  constexpr int NJ = TOK / 32;                      // 16-token sub-tiles per wave
  constexpr int W_IT = TAPS * TILE * 8 / 256;       // 16-byte weight units per thread per chunk
  constexpr bool CVT = (sizeof(T) == 2) && !XH;     // fp32 activations converted to bf16 while staging
  constexpr int XU = CVT ? 16 : 8;                  // 16-byte global units per activation row per chunk
  constexpr int XE = XH ? 8 : 4;                    // elements per unit
  constexpr int X_IT = (XROWS * XU + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ws = smem;
  unsigned char* Xs = smem + TAPS * TILE * 128;

  const int tiles_n = (a.N + TOK - 1) / TOK;
  const int b = blockIdx.x / tiles_n;
  const int n0 = (blockIdx.x - b * tiles_n) * TOK;
  const int co0 = blockIdx.y * TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wt = wave & 1;
  const int r = lane & 15, g = lane >> 4;
  const T* Wp = reinterpret_cast<const T*>(a.Wp);
  const int NL = a.rows_exist ? a.rows_exist[b] : a.N;     // rows of this batch row that exist as conv input

  if (a.skip_halo >= 0 && n0 >= a.lens[b] + a.skip_halo) {
    // whole tile is padding beyond the halo: nothing downstream reads it with a non-zero weight; keep it defined
    if (!a.accumulate) {
      for (int u = tid; u < TOK * 32; u += 256) {
        const int row = u >> 5, q = u & 31;
        const int n = n0 + row, co = co0 + q * 4;
        if (n < a.N && co < a.Cout) {
          if (a.y_bf16) {
            dx_h16* dst = reinterpret_cast<dx_h16*>(a.Y) + ((size_t)b * a.N + n) * a.ldy + co;
            *reinterpret_cast<uint2*>(dst) = make_uint2(0u, 0u);       // bf16 outputs always have Cout % 4 == 0
          } else {
            float* dst = a.Y + ((size_t)b * a.N + n) * a.ldy + co;
            if (co + 3 < a.Cout) *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
            else for (int e = 0; co + e < a.Cout; ++e) dst[e] = 0.f;
          }
        }
      }
    }
    return;
  }

  // prefetch registers; plain macros (not lambdas) so that the arrays are provably register-resident
  f32x4 wreg[W_IT], xreg[X_IT];                      // native vectors: SROA keeps them in VGPRs
#define DX_LOAD_CHUNK(CH)                                                                                                        \
  {                                                                                                                              \
    const int ci0_ = (CH) * BK;                                                                                                  \
    _Pragma("unroll") for (int it = 0; it < W_IT; ++it) {                                                                        \
      int row, q;                                                                                                                \
      w_unit<T>(tid + it * 256, row, q);                                                                                         \
      const int tap = row >> 7, col = row & (TILE - 1);                                                                          \
      if constexpr (sizeof(T) == 2) wreg[it] = *reinterpret_cast<const f32x4*>(Wp + wb_off(tap, co0 + col, ci0_ + q * 8, a.CoutP, a.CinP)); \
      else wreg[it] = *reinterpret_cast<const f32x4*>(Wp + ((size_t)(tap * a.CoutP + co0 + col) * a.CinP + ci0_) + q * 4);       \
    }                                                                                                                            \
    _Pragma("unroll") for (int it = 0; it < X_IT; ++it) {                                                                        \
      const int u = tid + it * 256;                                                                                              \
      const int row = u / XU, q = u % XU;                                                                                        \
      const int n = n0 + row - PAD, ci = ci0_ + q * XE;                                                                          \
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                                                       \
      if (u < XROWS * XU && n >= 0 && n < NL && ci < a.Cin) {                                                                    \
        if constexpr (XH) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const dx_h16*>(a.X) + ((size_t)b * a.N + n) * a.ldx + ci); \
        else v = *reinterpret_cast<const f32x4*>(a.X + ((size_t)b * a.N + n) * a.ldx + ci);                                      \
      }                                                                                                                          \
      xreg[it] = v;                                                                                                              \
    }                                                                                                                            \
  }
#define DX_STORE_CHUNK()                                                                                                         \
  {                                                                                                                              \
    _Pragma("unroll") for (int it = 0; it < W_IT; ++it) {                                                                        \
      int row, q;                                                                                                                \
      w_unit<T>(tid + it * 256, row, q);                                                                                         \
      *reinterpret_cast<f32x4*>(Ws + lds_off(row, q)) = wreg[it];                                                               \
    }                                                                                                                            \
    _Pragma("unroll") for (int it = 0; it < X_IT; ++it) {                                                                        \
      const int u = tid + it * 256;                                                                                              \
      const int row = u / XU, q = u % XU;                                                                                        \
      if (u < XROWS * XU) {                                                                                                      \
        if constexpr (CVT) *reinterpret_cast<uint2*>(Xs + lds_off(row, q >> 1) + ((q & 1) << 3)) = pack_bf16x4v(xreg[it]);       \
        else *reinterpret_cast<f32x4*>(Xs + lds_off(row, q)) = xreg[it];                                                         \
      }                                                                                                                          \
    }                                                                                                                            \
  }

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = a.CinP / BK;
  DX_LOAD_CHUNK(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();                       // every wave is done reading the previous chunk
    DX_STORE_CHUNK();
    __syncthreads();
    if (ch + 1 < nchunks) DX_LOAD_CHUNK(ch + 1);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float4 wf[4], xf[NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          wf[i] = *reinterpret_cast<const float4*>(Ws + lds_off(tap * TILE + wc * 64 + i * 16 + r, ks * 4 + g));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          xf[j] = *reinterpret_cast<const float4*>(Xs + lds_off(wt * (TOK / 2) + j * 16 + r + tap, ks * 4 + g));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) Mma<T>::run(wf[i], xf[j], acc[i][j]);
      }
    }
  }

  // epilogue: lane holds channels co..co+3 of token n
  const int len_b = a.lens ? a.lens[b] : a.N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + wc * 64 + i * 16 + g * 4;
    if (co >= a.Cout) continue;
    const bool full = (co + 3 < a.Cout);
    float bv[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (full) {                                          // one 16-byte load per epilogue vector instead of four scalar ones
      if (a.bias) { const float4 t = *reinterpret_cast<const float4*>(a.bias + co); bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w; }
      if (a.post_scale) {
        const float4 t = *reinterpret_cast<const float4*>(a.post_scale + co), u = *reinterpret_cast<const float4*>(a.post_shift + co);
        sc[0] = t.x; sc[1] = t.y; sc[2] = t.z; sc[3] = t.w; sh[0] = u.x; sh[1] = u.y; sh[2] = u.z; sh[3] = u.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (co + e < a.Cout) {
          if (a.bias) bv[e] = a.bias[co + e];
          if (a.post_scale) { sc[e] = a.post_scale[co + e]; sh[e] = a.post_shift[co + e]; }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = n0 + wt * (TOK / 2) + j * 16 + r;
      if (n >= a.N) continue;
      const size_t row = (size_t)b * a.N + n;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = acc[i][j][e] + bv[e];
        if (a.relu) t = fmaxf(t, 0.f);
        t = t * sc[e] + sh[e];
        v[e] = t * a.out_scale;
      }
      if (a.relu_aux) {
        if (a.aux_bf16) {
          const bf16x4 av = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const dx_h16*>(a.relu_aux) + row * a.ld_aux + co);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!((float)av[e] > 0.f)) v[e] = 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < a.Cout && !(a.relu_aux[row * a.ld_aux + co + e] > 0.f)) v[e] = 0.f;
        }
      }
      if (a.mask_rows && n >= len_b) { v[0] = v[1] = v[2] = v[3] = 0.f; }
      if (a.y_bf16) {
        dx_h16* dsth = reinterpret_cast<dx_h16*>(a.Y) + row * a.ldy + co;
        *reinterpret_cast<uint2*>(dsth) = pack_bf16x4(make_float4(v[0], v[1], v[2], v[3]));
        continue;
      }
      float* dst = a.Y + row * a.ldy + co;
      if (full) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (a.accumulate) {
          const float4 old = *reinterpret_cast<const float4*>(dst);
          o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *reinterpret_cast<float4*>(dst) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < a.Cout) dst[e] = a.accumulate ? dst[e] + v[e] : v[e];
      }
    }
  }
}

#undef DX_LOAD_CHUNK
#undef DX_STORE_CHUNK

// ------------------------------------------------------------------------------------------------
// Weight-stationary persistent variant for the short-K layers (Cin <= 128: MHA in/out projections, FF conv1 and the input
// gradient of FF conv2, mel projection, prenet conv0), bf16 operands.
// With K = TAPS*128 the tiled kernel above runs only two pipeline stages per workgroup, and ~60 % of a workgroup's life is
// fixed cost (first loads -> LDS -> fragments -> epilogue).  Here a 512-thread workgroup (8 waves, one per CU for TAPS=3)
// keeps its whole weight slice [2 chunks][TAPS][128 co][64 ci] (96 KB) in LDS for its lifetime and walks over token tiles
// (stride = workgroups per channel tile); the next tile's activations are prefetched into registers during the MFMAs.
// ------------------------------------------------------------------------------------------------
template <int TAPS, bool XH>
__global__ __launch_bounds__(512, 2) void conv_ws_kernel(const ConvGemmArgs a, int wgs_per_cotile) {
  constexpr int PAD = (TAPS - 1) / 2;
  constexpr int TOK = 128;
  constexpr int XROWS = TOK + TAPS - 1;
  constexpr int XU = XH ? 16 : 32;                         // 16-byte units per activation row (128 channels, bf16 or fp32)
  constexpr int XE = XH ? 8 : 4;
  constexpr int X_IT = (XROWS * XU + 511) / 512;
  constexpr int W_ROWS = 2 * TAPS * TILE;                  // [chunk][tap][co]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ws = smem;
  unsigned char* Xs = smem + W_ROWS * 128;                 // [chunk][XROWS] rows of 128 bytes; re-used as the output staging tile
  constexpr int OLD = 288;                                 // output staging row stride (bytes): 128 bf16 + pad
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wt = wave & 1;                 // 4 x 2 waves: 32 channels x 64 tokens each
  const int r = lane & 15, g = lane >> 4;
  const int co0 = blockIdx.y * TILE;
  const dx_h16* Wp = reinterpret_cast<const dx_h16*>(a.Wp);

  // weight slice -> LDS, once
  for (int u = tid; u < W_ROWS * 8; u += 512) {
    int row, q;
    w_unit<dx_h16>(u, row, q);
    const int ch = row / (TAPS * TILE), rem = row - ch * (TAPS * TILE);
    const int tap = rem >> 7, col = rem & (TILE - 1);
    *reinterpret_cast<f32x4*>(Ws + lds_off(row, q)) = *reinterpret_cast<const f32x4*>(Wp + wb_off(tap, co0 + col, ch * 64 + q * 8, a.CoutP, a.CinP));
  }

  const int tiles_n = (a.N + TOK - 1) / TOK;
  const int total = a.B * tiles_n;
  // Token tiles this workgroup walks.  With tile skipping on, the batch's LIVE tiles are numbered first (exclusive prefix of
  // live tiles per batch row in LDS, binary search per tile) and shared round-robin; padding tiles are zero-filled up front.
  // Walking (b, tile) in plain order with a stride gave some workgroups two live tiles and others none on the Cout = 128 layers
  // (256 workgroups, ~250 live tiles of 336): the launch took two tile times instead of one.
  __shared__ int pre_s[DK_MAX_B + 1];
  // (only when a workgroup gets few tiles: with ~8 tiles each the strided plain order is balanced enough and skips the search)
  const bool compact = a.skip_halo >= 0 && a.B <= DK_MAX_B && total <= 4 * wgs_per_cotile;
  if (compact && wave == 0) {
    int run = 0;
    for (int base = 0; base < a.B; base += 64) {
      const int i = base + lane;
      const int cnt = i < a.B ? min(tiles_n, max(0, (min(a.lens[i] + a.skip_halo, a.N) + TOK - 1) / TOK)) : 0;
      int inc = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (lane >= off) inc += v; }
      if (i < a.B) pre_s[i] = run + inc - cnt;
      run += __shfl(inc, 63, 64);
    }
    if (lane == 0) pre_s[a.B] = run;
  }
  // plain order: the same array holds the per-row skip limits (a scalar global load per tile costs ~1 k cycles)
  const bool limits_in_lds = !compact && a.skip_halo >= 0 && a.B <= DK_MAX_B;
  if (limits_in_lds) for (int i = tid; i < a.B; i += 512) pre_s[i] = a.lens[i] + a.skip_halo;
  __syncthreads();
  const int nlive = compact ? pre_s[a.B] : total;
  // idx-th live (or padding) tile -> (b, n0)
  auto locate = [&](int idx, bool want_live, int& b, int& n0) {
    int lo = 0, hi = a.B;                                   // largest row lo with key(lo) <= idx
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      const int key = want_live ? pre_s[mid] : mid * tiles_n - pre_s[mid];
      if (key <= idx) lo = mid; else hi = mid;
    }
    b = lo;
    const int live_b = pre_s[b + 1] - pre_s[b];
    n0 = (want_live ? idx - pre_s[b] : live_b + (idx - (b * tiles_n - pre_s[b]))) * TOK;
  };
  auto zero_tile = [&](int b, int n0) {
    for (int u = tid; u < TOK * 32; u += 512) {
      const int row = u >> 5, q = u & 31;
      const int n = n0 + row, co = co0 + q * 4;
      if (n < a.N && co < a.Cout) {
        if (a.y_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<dx_h16*>(a.Y) + ((size_t)b * a.N + n) * a.ldy + co) = make_uint2(0u, 0u);
        else {
          float* dst = a.Y + ((size_t)b * a.N + n) * a.ldy + co;
          if (co + 3 < a.Cout) *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
          else for (int e = 0; co + e < a.Cout; ++e) dst[e] = 0.f;
        }
      }
    }
  };
  if (compact && !a.accumulate) {
    for (int d = blockIdx.x; d < total - nlive; d += wgs_per_cotile) {
      int zb, zn0;
      locate(d, false, zb, zn0);
      zero_tile(zb, zn0);
    }
  }
  f32x4 xreg[X_IT];
#define DX_WS_LOAD(B_, N0_)                                                                                          \
  const int nl_ = a.rows_exist ? a.rows_exist[B_] : a.N;                                                             \
  _Pragma("unroll") for (int it = 0; it < X_IT; ++it) {                                                              \
    const int u = tid + it * 512;                                                                                    \
    const int row = u / XU, q = u % XU;                                                                              \
    const int n = (N0_) + row - PAD, ci = q * XE;                                                                    \
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                                             \
    if (u < XROWS * XU && n >= 0 && n < nl_ && ci < a.Cin) {                                                         \
      if constexpr (XH) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const dx_h16*>(a.X) + ((size_t)(B_) * a.N + n) * a.ldx + ci); \
      else v = *reinterpret_cast<const f32x4*>(a.X + ((size_t)(B_) * a.N + n) * a.ldx + ci);                         \
    }                                                                                                                \
    xreg[it] = v;                                                                                                    \
  }
  // this workgroup's first WS_LIST live tiles are located up front, one per thread (a binary search per tile inside the loop sat
  // on the path to the next tile's prefetch)
  constexpr int WS_LIST = 128;
  __shared__ int mine_s[WS_LIST];                          // (b << 16) | tile index
  if (compact && tid < WS_LIST) {
    const int idx = blockIdx.x + tid * wgs_per_cotile;
    if (idx < nlive) {
      int lb, ln0;
      locate(idx, true, lb, ln0);
      mine_s[tid] = (lb << 16) | (ln0 / TOK);
    }
  }
  __syncthreads();
  int tile_no = 0;
  // next live tile of this workgroup (plain order: padding tiles beyond the halo are zero-filled on the way)
  auto next_tile = [&](int& t, int& b, int& n0) {
    if (compact) {
      if (t >= nlive) return false;
      if (tile_no < WS_LIST) { const int v = mine_s[tile_no]; b = v >> 16; n0 = (v & 0xffff) * TOK; }
      else locate(t, true, b, n0);
      ++tile_no;
      return true;
    }
    for (; t < total; t += wgs_per_cotile) {
      b = t / tiles_n;
      n0 = (t - b * tiles_n) * TOK;
      if (a.skip_halo < 0 || n0 < (limits_in_lds ? pre_s[b] : a.lens[b] + a.skip_halo)) return true;
      if (!a.accumulate) zero_tile(b, n0);
    }
    return false;
  };

  int t = blockIdx.x, b = 0, n0 = 0;
  bool live = next_tile(t, b, n0);
  if (live) { DX_WS_LOAD(b, n0) }
  while (live) {
    const int cb = b, cn0 = n0;
    __syncthreads();                                       // previous tile's fragment reads are done (and W is in place)
#pragma unroll
    for (int it = 0; it < X_IT; ++it) {
      const int u = tid + it * 512;
      const int row = u / XU, q = u % XU;                  // fp32: q = float4 index 0..31 -> chunk q>>4, 8-byte slot (q&15); bf16: q = 16-byte unit
      if (u < XROWS * XU) {
        if constexpr (XH) *reinterpret_cast<f32x4*>(Xs + (q >> 3) * (XROWS * 128) + lds_off(row, q & 7)) = xreg[it];
        else *reinterpret_cast<uint2*>(Xs + (q >> 4) * (XROWS * 128) + lds_off(row, (q & 15) >> 1) + ((q & 1) << 3)) = pack_bf16x4v(xreg[it]);
      }
    }
    __syncthreads();
    t += wgs_per_cotile;
    live = next_tile(t, b, n0);
    if (live) { DX_WS_LOAD(b, n0) }
    // ReLU-gradient mask of THIS tile (input gradient of FF conv2: the bf16 hidden tensor): requested before the matrix block so
    // that the copy-out below does not start with a dependent global load
    // (k = 3 only: the k = 1 instantiations would lose a wave per SIMD to the 16 extra registers)
    bf16x8 auxreg[TAPS == 3 ? 4 : 1];
    const bool aux_pf = TAPS == 3 && a.relu_aux != nullptr && a.aux_bf16 && a.y_bf16;
    if constexpr (TAPS == 3) if (aux_pf) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int u = tid + it * 512;
        const int n = cn0 + (u >> 4), co = co0 + (u & 15) * 8;
        bf16x8 v = bf16x8{};
        if (n < a.N && co < a.Cout) v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const dx_h16*>(a.relu_aux) + ((size_t)cb * a.N + n) * a.ld_aux + co);
        auxreg[it] = v;
      }
    }

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          float4 wf[2], xf[4];
#pragma unroll
          for (int i = 0; i < 2; ++i)
            wf[i] = *reinterpret_cast<const float4*>(Ws + lds_off((ch * TAPS + tap) * TILE + wc * 32 + i * 16 + r, ks * 4 + g));
#pragma unroll
          for (int j = 0; j < 4; ++j)
            xf[j] = *reinterpret_cast<const float4*>(Xs + ch * (XROWS * 128) + lds_off(wt * 64 + j * 16 + r + tap, ks * 4 + g));
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<dx_h16>::run(wf[i], xf[j], acc[i][j]);
        }

    const int len_b = a.lens ? a.lens[cb] : a.N;
    if (a.y_bf16) {
      // bf16 outputs (the 1024-wide hidden tensors): row-per-lane 8-byte global stores are store-issue bound, so the tile goes
      // through LDS and leaves as fully coalesced 16-byte row segments; ReLU-mask / padding mask are applied on the way out
      __syncthreads();                                     // every wave has finished reading the activation tile
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cl = wc * 32 + i * 16 + g * 4;
        const int co = co0 + cl;
        float bv[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (co < a.Cout) {                                 // bf16 outputs have Cout % 4 == 0
          if (a.bias) { const float4 tb = *reinterpret_cast<const float4*>(a.bias + co); bv[0] = tb.x; bv[1] = tb.y; bv[2] = tb.z; bv[3] = tb.w; }
          if (a.post_scale) {
            const float4 ts = *reinterpret_cast<const float4*>(a.post_scale + co), tsh = *reinterpret_cast<const float4*>(a.post_shift + co);
            sc[0] = ts.x; sc[1] = ts.y; sc[2] = ts.z; sc[3] = ts.w; sh[0] = tsh.x; sh[1] = tsh.y; sh[2] = tsh.z; sh[3] = tsh.w;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float tv = acc[i][j][e] + bv[e];
            if (a.relu) tv = fmaxf(tv, 0.f);
            v[e] = (tv * sc[e] + sh[e]) * a.out_scale;
          }
          *reinterpret_cast<uint2*>(Xs + (wt * 64 + j * 16 + r) * OLD + cl * 2) = pack_bf16x4(make_float4(v[0], v[1], v[2], v[3]));
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int u = tid + it * 512;
        const int row = u >> 4, q = u & 15;
        const int n = cn0 + row, co = co0 + q * 8;
        if (n < a.N && co < a.Cout) {
          bf16x8 o = *reinterpret_cast<const bf16x8*>(Xs + row * OLD + q * 16);
          const size_t grow = (size_t)cb * a.N + n;
          if (a.relu_aux) {
            if (a.aux_bf16) {
              bf16x8 av;
              if constexpr (TAPS == 3) av = auxreg[it];
              else av = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const dx_h16*>(a.relu_aux) + grow * a.ld_aux + co);
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (!((float)av[e] > 0.f)) o[e] = (dx_h16)0.f;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (!(a.relu_aux[grow * a.ld_aux + co + e] > 0.f)) o[e] = (dx_h16)0.f;
            }
          }
          if (a.mask_rows && n >= len_b) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (dx_h16)0.f;
          }
          if (co + 7 < a.Cout) *reinterpret_cast<bf16x8*>(reinterpret_cast<dx_h16*>(a.Y) + grow * a.ldy + co) = o;
          else *reinterpret_cast<bf16x4*>(reinterpret_cast<dx_h16*>(a.Y) + grow * a.ldy + co) = bf16x4{o[0], o[1], o[2], o[3]};
        }
      }
      continue;                                            // the loop-top barrier orders these LDS reads before the next tile's stores
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int co = co0 + wc * 32 + i * 16 + g * 4;
      if (co >= a.Cout) continue;
      const bool full = (co + 3 < a.Cout);
      float bv[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (co + e < a.Cout) {
          if (a.bias) bv[e] = a.bias[co + e];
          if (a.post_scale) { sc[e] = a.post_scale[co + e]; sh[e] = a.post_shift[co + e]; }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = cn0 + wt * 64 + j * 16 + r;
        if (n >= a.N) continue;
        const size_t row = (size_t)cb * a.N + n;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float tv = acc[i][j][e] + bv[e];
          if (a.relu) tv = fmaxf(tv, 0.f);
          tv = tv * sc[e] + sh[e];
          v[e] = tv * a.out_scale;
        }
        if (a.relu_aux) {
          if (a.aux_bf16) {
            const bf16x4 av = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const dx_h16*>(a.relu_aux) + row * a.ld_aux + co);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (!((float)av[e] > 0.f)) v[e] = 0.f;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (co + e < a.Cout && !(a.relu_aux[row * a.ld_aux + co + e] > 0.f)) v[e] = 0.f;
          }
        }
        if (a.mask_rows && n >= len_b) { v[0] = v[1] = v[2] = v[3] = 0.f; }
        if (a.y_bf16) {
          *reinterpret_cast<uint2*>(reinterpret_cast<dx_h16*>(a.Y) + row * a.ldy + co) = pack_bf16x4(make_float4(v[0], v[1], v[2], v[3]));
          continue;
        }
        float* dst = a.Y + row * a.ldy + co;
        if (full) {
          float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (a.accumulate) {
            const float4 old = *reinterpret_cast<const float4*>(dst);
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
          }
          *reinterpret_cast<float4*>(dst) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < a.Cout) dst[e] = a.accumulate ? dst[e] + v[e] : v[e];
        }
      }
    }
  }
#undef DX_WS_LOAD
}

template <int TAPS, bool XH>
void launch_conv_ws(const ConvGemmArgs& a, hipStream_t s) {
  const size_t smem = (size_t)(2 * TAPS * TILE) * 128 + std::max<size_t>((size_t)2 * (128 + TAPS - 1) * 128, (size_t)128 * 288);
  static bool configured = false;
  if (!configured) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ws_kernel<TAPS, XH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  const int co_tiles = a.CoutP / TILE;
  const int total = a.B * dx_cdiv(a.N, 128);
  const int per_cu = smem * 2 <= 160 * 1024 ? 2 : 1;       // TAPS=1 fits twice per CU
  const int wgs = std::max(1, std::min(total, (256 * per_cu) / co_tiles));
  hipLaunchKernelGGL((conv_ws_kernel<TAPS, XH>), dim3(wgs, co_tiles), dim3(512), smem, s, a, wgs);
}

// ------------------------------------------------------------------------------------------------
// Deep-K variant (bf16 operands, CinP >= 256: FF conv2 1024 -> 128, the input gradient of FF conv1, the 1024-wide prenet
// convs).  With Cout = 128 these launches have at most ~one workgroup per CU and a 16-48 stage K loop, so they run at the
// latency of ONE workgroup's loop, and in the tiled kernel that loop is dominated by moving the weight tile: 49 of the 65 KB
// staged per 64-deep stage are weights, written to LDS at ~79 B/clk (13 cycles per ds_write_b128) and read back as fragments.
// Here the weights never touch LDS: the bf16 pack is fragment-major (wb_off), so the 16 bytes a lane needs for one 16x32 MFMA
// "A" fragment are contiguous and a wave fetches a fragment with ONE coalesced 1 KB load straight into registers, one stage
// ahead (L2 resident: every workgroup reads the same slice).  Only the activation tile (16.6 KB per stage, two LDS buffers,
// one barrier per stage) goes through LDS; its loads run two stages ahead (they come from HBM / MALL).
// 512 threads: the 64-deep stage is split in K between two groups of four waves, each wave owning 32 channels x 128 tokens,
// so no two waves fetch the same weight fragment, every SIMD holds two waves, and LDS fragment traffic is what a 64x64 wave
// tile would read.  The two K halves are summed through LDS at the end and each group finishes half of the tokens.
// ------------------------------------------------------------------------------------------------
template <int TAPS, bool XH>
__global__ __launch_bounds__(512) void conv_dk_kernel(const ConvGemmArgs a) {
  constexpr int PAD = (TAPS - 1) / 2;
  constexpr int TOK = 128;
  constexpr int XROWS = TOK + TAPS - 1;
  constexpr int XU = XH ? 8 : 16;                   // 16-byte global units per activation row per stage
  constexpr int XE = XH ? 8 : 4;
  constexpr int X_IT = (XROWS * XU + 511) / 512;
  constexpr int XBUF = XROWS * 128;
  constexpr int NW = TAPS * 2;                      // weight fragments per wave per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tiles_n = (a.N + TOK - 1) / TOK;
  // Workgroup -> (token tile, output-channel block).  Workgroups are dispatched in linear order (x fastest) and dealt to the 8 XCDs
  // round-robin.  With the plain (tile, block) grid all tiles of block 0 run first, then all of block 1 ...: every activation tile is
  // fetched once per block, a round of workgroups (~70 MB of activations) apart, i.e. from HBM each time (PMC: 3.2x the algorithmic
  // bytes for the 1024 -> 1024 layers).  Renumbered: XCD x owns the token tiles t = 8 i + x, and on that XCD the workgroups of a
  // tile's blocks are CONSECUTIVE, in groups of at most four blocks (4 x 786 KB of weights stay L2-resident; all eight would not fit
  // the 4 MB next to the activations): the tile is read from HBM once per group and served to the other blocks from the XCD's L2.
  int tile_id = blockIdx.x, co_blk = blockIdx.y;
  if (a.xcd_swizzle) {
    const int nco = gridDim.y, tpx = gridDim.x >> 3;                  // (the host pads gridDim.x to a multiple of 8)
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    const int cg = nco < a.xcd_swizzle ? nco : a.xcd_swizzle;      // (xcd_swizzle = blocks per group: 4)
    const int per_group = tpx * cg;
    const int grp = j / per_group, rem = j - grp * per_group;
    const int t = rem / cg;
    tile_id = t * 8 + xcd;
    co_blk = grp * cg + (rem - t * cg);
    if (tile_id >= a.B * tiles_n) return;                              // padding of the grid
  }
  const int co0 = co_blk * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: everything per-wave below stays in SGPRs
  const int kg = wave >> 2, wq = wave & 3;          // waves w and w+4 share a SIMD: one of each K group
  const int r = lane & 15, g = lane >> 4;
  const dx_h16* Wp = reinterpret_cast<const dx_h16*>(a.Wp);

  // Workgroup -> token tile.  A workgroup fills a CU (registers), so a launch with about as many live tiles as CUs must not
  // lose a CU to a padding tile: with tile skipping on, the live tiles of the batch are numbered first (workgroups go to the
  // XCDs round-robin, so consecutive ids spread evenly) and the workgroups after them zero-fill the padding tiles.
  __shared__ int pre_s[DK_MAX_B + 1];               // exclusive prefix of live tiles per batch row, [B] = total
  int b, n0;
  bool live = true;
  if (a.skip_halo >= 0 && a.B <= DK_MAX_B) {
    if (wave == 0) {
      int run = 0;
      for (int base = 0; base < a.B; base += 64) {
        const int i = base + lane;
        const int cnt = i < a.B ? min(tiles_n, max(0, (min(a.lens[i] + a.skip_halo, a.N) + TOK - 1) / TOK)) : 0;
        int inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (lane >= off) inc += v; }
        if (i < a.B) pre_s[i] = run + inc - cnt;
        run += __shfl(inc, 63, 64);
      }
      if (lane == 0) pre_s[a.B] = run;
    }
    __syncthreads();
    const int nlive = pre_s[a.B];
    int w = tile_id, lo = 0, hi = a.B;                // largest row lo with key(lo) <= w, key = live (dead) tiles before the row
    live = w < nlive;
    if (!live) w -= nlive;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      const int key = live ? pre_s[mid] : mid * tiles_n - pre_s[mid];
      if (key <= w) lo = mid; else hi = mid;
    }
    b = lo;
    const int live_b = pre_s[b + 1] - pre_s[b];
    n0 = (live ? w - pre_s[b] : live_b + (w - (b * tiles_n - pre_s[b]))) * TOK;
  } else {
    b = tile_id / tiles_n;
    n0 = (tile_id - b * tiles_n) * TOK;
    if (a.skip_halo >= 0) live = n0 < a.lens[b] + a.skip_halo;
  }

  if (!live) {                                        // padding beyond the halo: keep the output defined
    if (!a.accumulate) {
      for (int u = tid; u < TOK * 32; u += 512) {
        const int row = u >> 5, q = u & 31;
        const int n = n0 + row, co = co0 + q * 4;
        if (n < a.N && co < a.Cout) {
          if (a.y_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<dx_h16*>(a.Y) + ((size_t)b * a.N + n) * a.ldy + co) = make_uint2(0u, 0u);
          else {
            float* dst = a.Y + ((size_t)b * a.N + n) * a.ldy + co;
            if (co + 3 < a.Cout) *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
            else for (int e = 0; co + e < a.Cout; ++e) dst[e] = 0.f;
          }
        }
      }
    }
    return;
  }

  const int nchunks = a.CinP / 64;
  const int NL = a.rows_exist ? a.rows_exist[b] : a.N;              // rows of this batch row that exist as conv input
  // Every staging access is unconditional (addresses are clamped, zeros are selected in when the value is stored): loads
  // inside divergent branches make the compiler give up counting vmcnt and drain the whole prefetch queue at each use.
  // Addresses are a wave-uniform base (scalar unit) + a 32-bit per-thread element offset: no vector arithmetic per load (the
  // host routes Cin % 64 != 0 or >= 2^31-element tensors to the tiled kernel).  Rows outside the batch row read row 0 of the
  // tensor instead and are zeroed at the LDS store, which only the first / last tile of a batch row ever needs.
  int xoff[X_IT], xlds[X_IT]; bool xpad[X_IT];
#pragma unroll
  for (int it = 0; it < X_IT; ++it) {
    const int u = min(tid + it * 512, XROWS * XU - 1);              // surplus slots repeat the last unit (same value, same place)
    const int row = u / XU, q = u % XU;
    const int n = n0 + row - PAD;
    xpad[it] = n < 0 || n >= NL;
    xoff[it] = xpad[it] ? q * XE : (b * a.N + n) * a.ldx + q * XE;
    xlds[it] = XH ? lds_off(row, q) : lds_off(row, q >> 1) + ((q & 1) << 3);
  }
  const bool edge_tile = n0 < PAD || n0 + TOK + PAD > NL;           // wave-uniform
  // fragment (tap, i) of stage ch: element offset wbase + ((tap * (CoutP/16) + i) * (CinP/32) + 2 * ch) * 512
  const dx_h16* const wwave = Wp + ((size_t)((co0 >> 4) + wq * 2) * (a.CinP >> 5) + kg) * 512;      // wave-uniform
  const size_t wtap = (size_t)(a.CoutP >> 4) * (a.CinP >> 5) * 512, wrow = (size_t)(a.CinP >> 5) * 512;
  const int wlane = lane * 8;
  f32x4 wa[NW], wb[NW], xa[X_IT], xb[XH ? X_IT : 1];
  // chunk indices past the end re-read the last chunk (never consumed)
#define DK_LOAD_W2(TAP, CH, WR)                                                                                      \
  {                                                                                                                  \
    const dx_h16* wp_ = wwave + (size_t)min((CH), nchunks - 1) * 1024 + (TAP) * wtap;                                \
    WR[(TAP) * 2] = *reinterpret_cast<const f32x4*>(wp_ + wlane);                                                    \
    WR[(TAP) * 2 + 1] = *reinterpret_cast<const f32x4*>(wp_ + wrow + wlane);                                         \
  }
#define DK_LOAD_W(CH, WR) { _Pragma("unroll") for (int tap = 0; tap < TAPS; ++tap) DK_LOAD_W2(tap, CH, WR) }
#define DK_LOAD_X1(IT, CH, XR)                                                                                       \
  {                                                                                                                  \
    const int ch_ = min((CH), nchunks - 1) * 64;                                                                     \
    if constexpr (XH) XR[IT] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const dx_h16*>(a.X) + ch_ + xoff[IT]); \
    else XR[IT] = *reinterpret_cast<const f32x4*>(a.X + ch_ + xoff[IT]);                                             \
  }
#define DK_LOAD_X(CH, XR) { _Pragma("unroll") for (int it = 0; it < X_IT; ++it) DK_LOAD_X1(it, CH, XR) }
#define DK_STORE_X1(IT, CH, BASE, XR)                                                                                \
  {                                                                                                                  \
    f32x4 v_ = XR[IT];                                                                                               \
    if (edge_tile) v_ = xpad[IT] ? f32x4{0.f, 0.f, 0.f, 0.f} : v_;                                                   \
    if constexpr (XH) *reinterpret_cast<f32x4*>((BASE) + xlds[IT]) = v_;                                             \
    else *reinterpret_cast<uint2*>((BASE) + xlds[IT]) = pack_bf16x4v(v_);                                            \
  }
#define DK_STORE_X(CH, BASE, XR) { _Pragma("unroll") for (int it = 0; it < X_IT; ++it) DK_STORE_X1(it, CH, BASE, XR) }
  // accumulators as sixteen named vectors (c<i><j>: 16-channel sub-tile i, 16-token sub-tile j) and the matrix block written
  // out per tap: no indexed arrays, so every copy of the block keeps its state in VGPRs
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 c00 = zero4, c01 = zero4, c02 = zero4, c03 = zero4, c04 = zero4, c05 = zero4, c06 = zero4, c07 = zero4;
  f32x4 c10 = zero4, c11 = zero4, c12 = zero4, c13 = zero4, c14 = zero4, c15 = zero4, c16 = zero4, c17 = zero4;
  const int xfrag0 = lds_off(r, kg * 4 + g);                        // + j * 16 * 128
  const int xfrag1 = lds_off(r + 1, kg * 4 + g);
  const int xfrag2 = lds_off(r + 2, kg * 4 + g);
#define DK_FRAG(P) (*reinterpret_cast<const float4*>(P))
#define DK_MMA(W, X, C) C = DX_MFMA_H16(__builtin_bit_cast(bf16x8, W), __builtin_bit_cast(bf16x8, X), C);
  // One stage of the K loop.  Stamps of the plain order (stage activations, request loads, 48 MFMAs per wave, barrier) gave
  // 2.8 k cycles per stage: 1.75 k with both waves of a SIMD in their MFMAs (18 cycles each) and 1.05 k with both of them
  // stuck ISSUING memory instructions (the CU's address path takes a 1 KB wave-load per 16 cycles: 72 loads = 1.15 k cycles
  // per stage) while the matrix pipe idles.  A wave issues in order, so the only way to overlap the two is inside the wave:
  // the stage is cut into half-taps of 8 MFMAs (128 cycles of matrix pipe) and its memory work is spread between them -
  // fragment reads one half-tap ahead (two register sets), the LDS store + global request of one activation unit, or the
  // refill of one tap's weight registers (free once that tap's MFMAs have issued), per gap.  sched_barriers pin the order.
  float4 p0, p1, p2, p3, q0, q1, q2, q3;
#define DK_RD4(P, X0, X1, X2, X3) { const unsigned char* xp_ = (P); X0 = DK_FRAG(xp_); X1 = DK_FRAG(xp_ + 2048); X2 = DK_FRAG(xp_ + 4096); X3 = DK_FRAG(xp_ + 6144); }
#define DK_MM8(W0, W1, X0, X1, X2, X3, A0, A1, A2, A3, B0, B1, B2, B3)                                               \
  {                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    DK_MMA(W0, X0, A0) DK_MMA(W0, X1, A1) DK_MMA(W0, X2, A2) DK_MMA(W0, X3, A3)                                       \
    DK_MMA(W1, X0, B0) DK_MMA(W1, X1, B1) DK_MMA(W1, X2, B2) DK_MMA(W1, X3, B3)                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
#define DK_MM_LO(W0, W1, X0, X1, X2, X3) DK_MM8(W0, W1, X0, X1, X2, X3, c00, c01, c02, c03, c10, c11, c12, c13)
#define DK_MM_HI(W0, W1, X0, X1, X2, X3) DK_MM8(W0, W1, X0, X1, X2, X3, c04, c05, c06, c07, c14, c15, c16, c17)
  // activation unit IT of the stage: chunk C+1 (held in XS since two stages ago) -> buffer NXT, then request chunk C+3 into
  // the same registers (fp32 rows: one register set, one stage ahead)
#define DK_XSLOT(IT, C, NXT, XS)                                                                                     \
  if constexpr ((IT) < X_IT) {                                                                                       \
    if constexpr (XH) { DK_STORE_X1(IT, (C) + 1, NXT, XS) DK_LOAD_X1(IT, (C) + 3, XS) }                              \
    else { DK_STORE_X1(IT, (C) + 1, NXT, xa) DK_LOAD_X1(IT, (C) + 2, xa) }                                           \
  }
  // stage C: this wave's half (kg) of the K range of chunk C from buffer CUR / weight registers WR, all taps
#define DK_STAGE(C, CUR, NXT, XS, WR)                                                                                \
  {                                                                                                                  \
    DK_RD4((CUR) + xfrag0, p0, p1, p2, p3)                                                                           \
    DK_RD4((CUR) + xfrag0 + 8192, q0, q1, q2, q3)                                                                    \
    DK_MM_LO(WR[0], WR[1], p0, p1, p2, p3)                                                                           \
    if constexpr (TAPS == 3) {                                                                                       \
      DK_RD4((CUR) + xfrag1, p0, p1, p2, p3)                                                                         \
      DK_XSLOT(0, C, NXT, XS) DK_XSLOT(3, C, NXT, XS)                                                                \
      DK_MM_HI(WR[0], WR[1], q0, q1, q2, q3)                                                                         \
      DK_RD4((CUR) + xfrag1 + 8192, q0, q1, q2, q3)                                                                  \
      DK_LOAD_W2(0, (C) + 2, WR)                                                                                     \
      DK_MM_LO(WR[2], WR[3], p0, p1, p2, p3)                                                                         \
      DK_RD4((CUR) + xfrag2, p0, p1, p2, p3)                                                                         \
      DK_XSLOT(1, C, NXT, XS) DK_XSLOT(4, C, NXT, XS)                                                                \
      DK_MM_HI(WR[2], WR[3], q0, q1, q2, q3)                                                                         \
      DK_RD4((CUR) + xfrag2 + 8192, q0, q1, q2, q3)                                                                  \
      DK_LOAD_W2(1, (C) + 2, WR)                                                                                     \
      DK_MM_LO(WR[4], WR[5], p0, p1, p2, p3)                                                                         \
      DK_XSLOT(2, C, NXT, XS)                                                                                        \
      DK_MM_HI(WR[4], WR[5], q0, q1, q2, q3)                                                                         \
      DK_LOAD_W2(2, (C) + 2, WR)                                                                                     \
    } else {                                                                                                         \
      DK_XSLOT(0, C, NXT, XS) DK_XSLOT(1, C, NXT, XS)                                                                \
      DK_MM_HI(WR[0], WR[1], q0, q1, q2, q3)                                                                         \
      DK_XSLOT(2, C, NXT, XS) DK_XSLOT(3, C, NXT, XS)                                                                \
      DK_LOAD_W2(0, (C) + 2, WR)                                                                                     \
    }                                                                                                                \
    __syncthreads();                                                                                                 \
  }
  // the last chunk of an odd stage count: nothing left to stage
#define DK_MFMA_ONLY(CUR, WR)                                                                                        \
  {                                                                                                                  \
    DK_RD4((CUR) + xfrag0, p0, p1, p2, p3)                                                                           \
    DK_RD4((CUR) + xfrag0 + 8192, q0, q1, q2, q3)                                                                    \
    DK_MM_LO(WR[0], WR[1], p0, p1, p2, p3)                                                                           \
    DK_MM_HI(WR[0], WR[1], q0, q1, q2, q3)                                                                           \
    if constexpr (TAPS == 3) {                                                                                       \
      DK_RD4((CUR) + xfrag1, p0, p1, p2, p3)                                                                         \
      DK_RD4((CUR) + xfrag1 + 8192, q0, q1, q2, q3)                                                                  \
      DK_MM_LO(WR[2], WR[3], p0, p1, p2, p3)                                                                         \
      DK_MM_HI(WR[2], WR[3], q0, q1, q2, q3)                                                                         \
      DK_RD4((CUR) + xfrag2, p0, p1, p2, p3)                                                                         \
      DK_RD4((CUR) + xfrag2 + 8192, q0, q1, q2, q3)                                                                  \
      DK_MM_LO(WR[4], WR[5], p0, p1, p2, p3)                                                                         \
      DK_MM_HI(WR[4], WR[5], q0, q1, q2, q3)                                                                         \
    }                                                                                                                \
  }
  static_assert(X_IT <= 5, "activation units per thread per stage");
  unsigned char* const buf0 = smem;
  unsigned char* const buf1 = smem + XBUF;
  DK_LOAD_X(0, xa)
  DK_LOAD_W(0, wa)
  DK_STORE_X(0, buf0, xa)
  // issue order = roughly the order the loop leaves behind, pinned so that the vmcnt counts merged at the loop header are
  // close to the steady-state ones
  if constexpr (XH) {
    DK_LOAD_X(1, xb)
    __builtin_amdgcn_sched_barrier(0);
    DK_LOAD_X(2, xa)
  } else {
    DK_LOAD_X(1, xa)
  }
  __builtin_amdgcn_sched_barrier(0);
  DK_LOAD_W(1, wb)
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  // stages go in unconditional pairs plus a tail: a skippable second half would put a path into the loop on which the
  // youngest loads are different ones, and every wait would be sized for that path
  for (int c = 0; c + 1 < nchunks; c += 2) {
    DK_STAGE(c, buf0, buf1, xb, wa)
    DK_STAGE(c + 1, buf1, buf0, xa, wb)
  }
  if (nchunks & 1) {                                                  // odd stage count: the last chunk sits in buffer 0 / wa
    DK_MFMA_ONLY(buf0, wa)
    __syncthreads();
  }
#undef DK_LOAD_W2
#undef DK_LOAD_W
#undef DK_LOAD_X1
#undef DK_LOAD_X
#undef DK_STORE_X1
#undef DK_STORE_X
#undef DK_RD4
#undef DK_MM8
#undef DK_MM_LO
#undef DK_MM_HI
#undef DK_XSLOT
#undef DK_STAGE
#undef DK_MFMA_ONLY
#undef DK_MMA
#undef DK_FRAG

  // sum the two K halves: each wave hands the token half it does not finish to its partner (wave ^ 4) through LDS.
  // The loop's last barrier has retired every fragment read, so the activation buffers are free.
  unsigned char* const ex = smem + (size_t)(wq * 2) * 8 * 1024;
#define DK_GIVE(SLOT, LO, HI) *reinterpret_cast<f32x4*>(ex + (kg * 8 + (SLOT)) * 1024 + lane * 16) = kg == 0 ? (HI) : (LO);
  DK_GIVE(0, c00, c04) DK_GIVE(1, c01, c05) DK_GIVE(2, c02, c06) DK_GIVE(3, c03, c07)
  DK_GIVE(4, c10, c14) DK_GIVE(5, c11, c15) DK_GIVE(6, c12, c16) DK_GIVE(7, c13, c17)
#undef DK_GIVE
  __syncthreads();
  f32x4 res[2][4];
#define DK_TAKE(I, JJ, LO, HI) res[I][JJ] = (kg == 0 ? (LO) : (HI)) + *reinterpret_cast<const f32x4*>(ex + ((1 - kg) * 8 + (I) * 4 + (JJ)) * 1024 + lane * 16);
  DK_TAKE(0, 0, c00, c04) DK_TAKE(0, 1, c01, c05) DK_TAKE(0, 2, c02, c06) DK_TAKE(0, 3, c03, c07)
  DK_TAKE(1, 0, c10, c14) DK_TAKE(1, 1, c11, c15) DK_TAKE(1, 2, c12, c16) DK_TAKE(1, 3, c13, c17)
#undef DK_TAKE

  // epilogue: lane holds channels co..co+3 of token n; this wave finishes token sub-tiles 4*kg .. 4*kg+3 of its 32 channels
  const int len_b = a.lens ? a.lens[b] : a.N;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int co = co0 + wq * 32 + i * 16 + g * 4;
    if (co >= a.Cout) continue;
    const bool full = (co + 3 < a.Cout);
    float bv[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (co + e < a.Cout) {
        if (a.bias) bv[e] = a.bias[co + e];
        if (a.post_scale) { sc[e] = a.post_scale[co + e]; sh[e] = a.post_shift[co + e]; }
      }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int n = n0 + (kg * 4 + jj) * 16 + r;
      if (n >= a.N) continue;
      const size_t row = (size_t)b * a.N + n;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = res[i][jj][e] + bv[e];
        if (a.relu) t = fmaxf(t, 0.f);
        t = t * sc[e] + sh[e];
        v[e] = t * a.out_scale;
      }
      if (a.relu_aux) {
        if (a.aux_bf16) {
          const bf16x4 av = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const dx_h16*>(a.relu_aux) + row * a.ld_aux + co);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!((float)av[e] > 0.f)) v[e] = 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < a.Cout && !(a.relu_aux[row * a.ld_aux + co + e] > 0.f)) v[e] = 0.f;
        }
      }
      if (a.mask_rows && n >= len_b) { v[0] = v[1] = v[2] = v[3] = 0.f; }
      if (a.y_bf16) {
        *reinterpret_cast<uint2*>(reinterpret_cast<dx_h16*>(a.Y) + row * a.ldy + co) = pack_bf16x4(make_float4(v[0], v[1], v[2], v[3]));
        continue;
      }
      float* dst = a.Y + row * a.ldy + co;
      if (full) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (a.accumulate) {
          const float4 old = *reinterpret_cast<const float4*>(dst);
          o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *reinterpret_cast<float4*>(dst) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < a.Cout) dst[e] = a.accumulate ? dst[e] + v[e] : v[e];
      }
    }
  }
}

template <int TAPS, bool XH>
void launch_conv_dk(const ConvGemmArgs& a, hipStream_t s) {
  const size_t smem = std::max<size_t>((size_t)2 * (128 + TAPS - 1) * 128, (size_t)128 * 528);
  static bool configured = false;
  if (!configured) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dk_kernel<TAPS, XH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  dim3 grid(a.B * dx_cdiv(a.N, 128), a.CoutP / TILE);
  ConvGemmArgs b = a;
  static const int swz_env = getenv("DX_DK_SWIZZLE") ? atoi(getenv("DX_DK_SWIZZLE")) : 1;
  b.xcd_swizzle = (swz_env && grid.y > 1 && (grid.y <= 4 || grid.y % 4 == 0)) ? (swz_env == 1 ? 4 : swz_env) : 0;   // DX_DK_SWIZZLE = 8: all blocks of a tile together (diagnostic)
  if (b.xcd_swizzle && grid.y % b.xcd_swizzle != 0 && grid.y > (unsigned)b.xcd_swizzle) b.xcd_swizzle = 4;
  if (b.xcd_swizzle) grid.x = dx_roundup(grid.x, 8);
  hipLaunchKernelGGL((conv_dk_kernel<TAPS, XH>), grid, dim3(512), smem, s, b);
}

// ------------------------------------------------------------------------------------------------
// weight gradient: G[tap][co][ci] += sum over tokens of dY[token][co] * X[token + tap - PAD][ci]
// Both operands are token-major in HBM, i.e. K-major; they are staged untransposed ([token][channel],
// row stride 132 floats) and fragments are gathered with four ds_read_b32 per operand (conflict free:
// lanes 0-15 read consecutive channels, lane groups sit 4 rows = 16 banks apart).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* dY; int ldy;
  const float* X; int ldx;
  float* G;
  int B, N, Cin, Cout, ksplit;
  const int* lens; int skip_halo;   // chunks starting at or beyond len_b + skip_halo carry a zero dY: skipped
  float* dbias;                     // optional: dbias[co] += column sums of dY (fused bias gradient)
};

constexpr int WG_BK = 32;     // tokens per K chunk
constexpr int WG_LD = 132;    // LDS row stride in floats

template <int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
  constexpr int PAD = (TAPS - 1) / 2;
  __shared__ __attribute__((aligned(16))) float Ds[WG_BK * WG_LD];
  __shared__ __attribute__((aligned(16))) float Xs[(WG_BK + 2) * WG_LD];
  const int ci_tiles = (a.Cin + TILE - 1) / TILE;
  const int co0 = (blockIdx.x / ci_tiles) * TILE;
  const int ci0 = (blockIdx.x % ci_tiles) * TILE;
  const int tap = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave >> 1, wt = wave & 1;
  const int r = lane & 15, g = lane >> 4;

  const int chunks_per_row = (a.N + WG_BK - 1) / WG_BK;
  const int total = a.B * chunks_per_row;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = a.dbias != nullptr && tap == 0 && ci0 == 0 && tid < TILE;
  float bsum = 0.f;

  for (int c = blockIdx.z; c < total; c += a.ksplit) {   // interleaved split: every slice sees a mix of utterance lengths
    const int b = c / chunks_per_row;
    const int nc = (c - b * chunks_per_row) * WG_BK;
    if (a.skip_halo >= 0 && nc >= a.lens[b] + a.skip_halo) continue;
    const float* dYb = a.dY + (size_t)b * a.N * a.ldy;
    const float* Xb = a.X + (size_t)b * a.N * a.ldx;
    __syncthreads();
    for (int u = tid; u < WG_BK * 32; u += 256) {
      const int row = u >> 5, q = u & 31;
      const int n = nc + row, co = co0 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < a.N && co < a.Cout) v = *reinterpret_cast<const float4*>(dYb + (size_t)n * a.ldy + co);
      *reinterpret_cast<float4*>(&Ds[row * WG_LD + q * 4]) = v;
    }
    for (int u = tid; u < (WG_BK + TAPS - 1) * 32; u += 256) {
      const int row = u >> 5, q = u & 31;
      const int n = nc + row - PAD, ci = ci0 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n >= 0 && n < a.N && ci < a.Cin) v = *reinterpret_cast<const float4*>(Xb + (size_t)n * a.ldx + ci);
      *reinterpret_cast<float4*>(&Xs[row * WG_LD + q * 4]) = v;
    }
    __syncthreads();
    if (do_bias) {
#pragma unroll 8
      for (int k = 0; k < WG_BK; ++k) bsum += Ds[k * WG_LD + tid];
    }
#pragma unroll
    for (int kg = 0; kg < WG_BK / 16; ++kg) {
      float df[4][4], xf[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) df[i][e] = Ds[(kg * 16 + g * 4 + e) * WG_LD + wc * 64 + i * 16 + r];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) xf[j][e] = Xs[(kg * 16 + g * 4 + e + tap) * WG_LD + wt * 64 + j * 16 + r];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(df[i][e], xf[j][e], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ci = ci0 + wt * 64 + j * 16 + r;
      if (ci >= a.Cin) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + wc * 64 + i * 16 + g * 4 + e;
        if (co < a.Cout) atomicAdd(&a.G[((size_t)co * a.Cin + ci) * TAPS + tap], acc[i][j][e]);   // the parameter's own (Cout, Cin, taps) layout
      }
    }
  if (do_bias && co0 + tid < a.Cout && bsum != 0.f) atomicAdd(&a.dbias[co0 + tid], bsum);
}

// ------------------------------------------------------------------------------------------------
// weight gradient, bf16 MFMA operands (fp32 accumulate).  dY and X tiles are staged UNTRANSPOSED as bf16
// [token][channel] (288-B rows) and every lane gathers its 8 K(=token)-consecutive values of one channel with two
// ds_read_b64_tr_b16 (hardware transposed read: a 16-lane group reads a 4-token x 16-channel block and each lane
// receives one channel column) -- no transposing store pass, no shuffles.
// ------------------------------------------------------------------------------------------------
// tokens per K chunk: 128 where the staging registers fit without spilling (both operands stored bf16, or k = 1) - half the
// barriers and chunk bookkeeping per MFMA: k = 3 launches -2.4 %, k = 1 launches -8 %; 64 for fp32-stored operands with k = 3
constexpr int wb_bk(int taps, bool dyh, bool xh) { return (taps == 1 || (dyh && xh)) ? 128 : 64; }
constexpr int WB_LD = 144;     // LDS row stride in bf16 elements (288 B)
// Rows 8 apart sit on the same banks (8 x 288 B = 9 x 256 B), and the two 16-lane groups of a half-wave read rows 8 apart: a 2-way conflict
// on every transposed read (tools/ubench/lds_read_patterns.hip: 152 vs 269 B/clk at 8 waves).  So the 64-channel halves of a row are swapped
// in every second group of eight rows: element (row, c) lives at column c ^ WB_X(row).
#define WB_X(ROW_) ((((ROW_) >> 3) & 1) << 6)
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgradBf16Args {
  const void* dY; int ldy; int dy_bf16;
  const void* X; int ldx; int x_bf16;
  float* G;
  int B, N, Cin, Cout, ksplit;
  const int* lens; int skip_halo;
  float* dbias;
};

// Up to WG_MAX_JOBS weight gradients of one kind (same taps, same operand storage) in ONE launch: blockIdx.y = job.  A single 128 -> 1024
// k = 3 layer is 22 GFLOP (9 us of MFMA time) for a 1.5 MB result: filling 256 CUs with ONE layer needs a 16-way token split, i.e. 25 MB
// of fp32 atomics per layer (~19 us at the memory-side atomic rate) and 128 x 64 tiles that each stage their own dY copy.  Eight layers
// per launch fill the chip with the wide (128 x 128, one workgroup per CU) tile at a 4-way split: a quarter of the atomics and half the
// staging per MFMA.  The descriptors travel as kernel arguments (no device-side descriptor table to keep alive or to copy).
constexpr int WG_MAX_JOBS = 32;
struct WgradJobDev {
  const void* dY; const void* X; float* G; float* dbias; const int* lens;
  int ldy, ldx, B, N, Cin, Cout, skip_halo, pad_;
};
struct WgradBatchArgs {
  WgradJobDev job[WG_MAX_JOBS];
  int ksplit;
  int xcd_group;   // 1: the tiles of one (job, token slice) pair run on ONE XCD (see the kernel); needs gridDim.x == 8 and gridDim.y * gridDim.z % 8 == 0
  unsigned long long* stamps;   // diagnostic builds (-DDX_WG_STAMPS, tools/wgrad_stamps.py) only: [workgroup][wave][8] s_memtime values / sums
};
#ifdef DX_WG_STAMPS
static unsigned long long* g_wgrad_stamps = nullptr;
extern "C" void dx_wgrad_set_stamps(unsigned long long* p) { g_wgrad_stamps = p; }
#define WG_NOW() __builtin_amdgcn_s_memtime()
#define WG_STAMP(K, V) { if ((tid & 63) == 0 && ba.stamps) ba.stamps[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (tid >> 6)) * 8 + (K)] = (V); }
#else
#define WG_NOW() 0ull
#define WG_STAMP(K, V) {}
#endif

// the same transposed read as inline asm: invisible to hipcc's wait-count pass (callers place `s_waitcnt lgkmcnt` themselves)
template <int OFF>
__device__ __forceinline__ s16x4 dx_tr16_b64_asm(unsigned lds_addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
  return v;
}

__device__ __forceinline__ bf16x8 tr_fragment(const dx_h16* tile, int row0, int col0, int lane) {
  // rows row0 + 8g + [0,8), columns col0 + [0,16): lane (r = lane & 15, g = lane >> 4) gets column r, rows 8g..8g+7
  const int li = lane & 15, g = lane >> 4, q = li >> 2, p = li & 3;
  const int rowA = row0 + 8 * g + q, rowB = rowA + 4;
  const dx_h16* a0 = tile + rowA * WB_LD + ((col0 + 4 * p) ^ WB_X(rowA));
  const dx_h16* a1 = tile + rowB * WB_LD + ((col0 + 4 * p) ^ WB_X(rowB));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

static __device__ uint4 dx_wgrad_zero_unit[4];      // the source of DMA rows that do not exist (zero-initialised, never written)

// CIW = waves along the input channels: 2 -> 256 threads, 128 co x 64 ci per workgroup, two workgroups per CU;
//                                      4 -> 512 threads, 128 co x 128 ci, ONE workgroup per CU (Cin % 128 == 0).
// Per-workgroup records of the narrow form on 256 CUs: 384 workgroups leave half the CUs with two (110-120 k cycles) and half
// with one (65-90 k, then idle), and two co-resident workgroups gain only 13 % over one (each stages its own copy of the same dY
// tile).  The wide form puts the same two waves per SIMD on every CU with ONE dY tile per chunk: half the staging per MFMA.
template <int TAPS, bool DYH, bool XH, int CIW>
__global__ __launch_bounds__(CIW * 128, CIW == 2 ? 2 : 1) void wgrad_bf16_kernel(const WgradBatchArgs ba) {
  // Workgroup -> (tile, job, token slice).  The 8 tiles of an FFT-block layer share one whole operand (conv1: the 128-channel X, conv2:
  // the 128-channel dY) and read it chunk by chunk at the same pace.  Dispatched in grid order they sit on 8 DIFFERENT XCDs (blockIdx.x is
  // the fastest index and workgroups go to the XCDs round-robin), so that operand leaves the memory side eight times (PMC: 2.4x the algorithmic
  // bytes).  Renumbered, XCD x runs the pairs p = 8 i + x with all 8 tiles of a pair: one fetch per pair, seven L2 hits.
  int tile_id = blockIdx.x, job_id = blockIdx.y, split_id = blockIdx.z;
  [[maybe_unused]] const unsigned long long t_start = WG_NOW();
  [[maybe_unused]] unsigned long long t_vm = 0, t_bar = 0, n_chunks = 0;
  if (ba.xcd_group) {
    const int lin = blockIdx.x + 8 * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = lin & 7, j = lin >> 3;
    const int pair = (j >> 3) * 8 + xcd;
    tile_id = j & 7;
    job_id = pair % (int)gridDim.y;
    split_id = pair / (int)gridDim.y;
  }
  WgradBf16Args a;
  {
    const WgradJobDev& j = ba.job[job_id];                  // kernel-argument memory, wave-uniform index: scalar loads
    a.dY = j.dY; a.ldy = j.ldy; a.dy_bf16 = DYH; a.X = j.X; a.ldx = j.ldx; a.x_bf16 = XH; a.G = j.G;
    a.B = j.B; a.N = j.N; a.Cin = j.Cin; a.Cout = j.Cout; a.ksplit = ba.ksplit; a.lens = j.lens; a.skip_halo = j.skip_halo; a.dbias = j.dbias;
  }
  if (tile_id >= ((a.Cout + TILE - 1) / TILE) * ((a.Cin + CIW * 32 - 1) / (CIW * 32))) return;   // a job with fewer tiles than the widest
  // wave: 64 co x 32 ci -> 4 x 2 x TAPS MFMA tiles; waves = 2 (co) x CIW (ci).
  // The dY tile and the (halo-extended) X tile are staged once per 64-token chunk and shared by the taps.
  constexpr int PAD = (TAPS - 1) / 2;
  // DMA form (wide tile, both operands stored 16-bit): the tiles go global -> LDS by LDS-DMA loads into a ring of four 64-token images
  constexpr bool DMA = CIW == 4 && DYH && XH;
  constexpr int WB_BK = DMA ? 64 : wb_bk(TAPS, DYH, XH);
  constexpr int NT = CIW * 128;
  constexpr int CI_T = CIW * 32;
  constexpr int XROWS = WB_BK + TAPS - 1;
  constexpr int DU = DYH ? 16 : 32, DE = DYH ? 8 : 4;     // 16-byte global units per dY row (128 channels), elements per unit
  constexpr int XU = XH ? CI_T / 8 : CI_T / 4, XE = XH ? 8 : 4;   // per X row (CI_T channels)
  constexpr int D_IT = WB_BK * DU / NT;
  constexpr int X_IT = (XROWS * XU + NT - 1) / NT;
  // ONE LDS array: the dY tile, the X tile (rows beyond the halo stay zero so every tr read is in bounds), and - after the chunk
  // loop - the fp32 staging tile of the epilogue
  // The wide form (one workgroup per CU) keeps TWO images of the pair of tiles and runs a software pipeline over them (see the chunk loop).
  constexpr bool PIPE = CIW == 4;
  constexpr int IMG = (2 * WB_BK + 8) * WB_LD;                     // elements per image: dY tile, X tile + 8 zero rows
  // DMA images: unpadded 256-byte rows (an LDS-DMA wave-instruction writes 1 KB = four rows, lane-linear), XOR-swizzled 16-byte units
  constexpr int DMA_NIMG = 4, DMA_XG = (WB_BK + TAPS - 1 + 3) / 4;  // X row groups of four per image (k = 3: 17, the last one half halo, half zeros)
  constexpr int DMA_IMGB = (WB_BK + 4 * DMA_XG + 4) * 256;         // bytes per image (the window reads reach 2 rows past the last group)
  constexpr int SMEM_BYTES = DMA ? DMA_NIMG * DMA_IMGB : IMG * 2;
  // ONE __shared__ object (a second one beside an LDS-DMA destination makes hipcc drain the DMA queue before LDS reads): the
  // per-utterance limits of the chunk walk live in its last 2 KB
  constexpr int SMEM_ELEMS = DMA ? DMA_NIMG * DMA_IMGB / 2 : (PIPE ? 2 : 1) * IMG;
  __shared__ __attribute__((aligned(16))) dx_h16 smem_all[SMEM_ELEMS + 1024];
  int* const limit_s = reinterpret_cast<int*>(smem_all + SMEM_ELEMS);
  dx_h16* const Ds = smem_all;
  dx_h16* const Xs = smem_all + WB_BK * WB_LD;
  const int ci_tiles = (a.Cin + CI_T - 1) / CI_T;
  const int co0 = (tile_id / ci_tiles) * TILE;
  const int ci0 = (tile_id % ci_tiles) * CI_T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave / CIW, wt = wave % CIW;
  const int r = lane & 15, g = lane >> 4;
  const int chunks_per_row = (a.N + WB_BK - 1) / WB_BK;
  const int total = a.B * chunks_per_row;

  f32x4 acc[TAPS][4][2];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (DMA) {                     // the rows behind the last X group of every image stay zero
    for (int u = tid; u < DMA_NIMG * 4 * 128; u += NT)
      reinterpret_cast<dx_h16*>(reinterpret_cast<char*>(smem_all) + (u / 512) * DMA_IMGB + (WB_BK + 4 * DMA_XG) * 256)[u % 512] = (dx_h16)0.f;
  } else {
    for (int u = tid; u < 8 * WB_LD; u += NT) {
      Xs[WB_BK * WB_LD + u] = (dx_h16)0.f;
      if constexpr (PIPE) Xs[IMG + WB_BK * WB_LD + u] = (dx_h16)0.f;
    }
  }

  // Fused bias gradient (column sums of the dY tile), done by the workgroups of input-channel tile 0 with ALL their threads
  // (thread = channel pair x one slice of the rows, 4-byte LDS reads).  As 128 threads x 64 two-byte reads it made those
  // workgroups' first two waves ~1.2 k cycles per chunk slower than everyone else: the kernel's tail.  (Spreading the sums
  // over all sibling workgroups instead was far worse: 100 k atomics on the same 128 addresses.)
  const bool do_bias = a.dbias != nullptr && ci0 == 0;
  float bsum0 = 0.f, bsum1 = 0.f;
  f32x4 dreg[D_IT], xreg[X_IT];
  // Staging addresses: (unit -> tile row, channel) is fixed per thread, only the chunk's first row moves: a wave-uniform 64-bit
  // chunk base (scalar unit) plus a 32-bit per-thread constant.
  // A thread's units keep their channel from one iteration to the next (NT % DU == 0): the tile row advances by NT / DU, so ONE row, ONE
  // 32-bit offset per operand and scalar multiples of the leading dimension are all the addressing state (18 registers as arrays).
  static_assert(NT % DU == 0 && NT % XU == 0, "staging units");
  constexpr int DSTEP = NT / DU, XSTEP = NT / XU;
  const int dch = co0 + (tid % DU) * DE, xch = ci0 + (tid % XU) * XE;
  const int drow0 = dch < a.Cout ? tid / DU : 0x40000000;            // channel out of range: never in bounds
  const int xrow0 = xch < a.Cin ? tid / XU : 0x40000000;
  const int dtoff0 = (tid / DU) * a.ldy + dch, xtoff0 = (tid / XU) * a.ldx + xch;
#define DX_WG_LOAD(B_, NC_)                                                                                                   \
  {                                                                                                                           \
    const ptrdiff_t dbase_ = ((ptrdiff_t)(B_) * a.N + (NC_)) * a.ldy;                                                         \
    const ptrdiff_t xbase_ = ((ptrdiff_t)(B_) * a.N + (NC_) - PAD) * a.ldx;                                                   \
    const int dlim_ = a.N - (NC_), xlo_ = PAD - (NC_), xhi_ = a.N - (NC_) + PAD;                                              \
    _Pragma("unroll") for (int it = 0; it < D_IT; ++it) {                                                                     \
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                                                    \
      if (drow0 + it * DSTEP < dlim_) {                                                                                       \
        const ptrdiff_t o_ = dbase_ + (ptrdiff_t)(it * DSTEP) * a.ldy;                                                        \
        if constexpr (DYH) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const dx_h16*>(a.dY) + o_ + dtoff0);          \
        else v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.dY) + o_ + dtoff0);                         \
      }                                                                                                                       \
      dreg[it] = v;                                                                                                           \
    }                                                                                                                         \
    _Pragma("unroll") for (int it = 0; it < X_IT; ++it) {                                                                     \
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};                                                                                    \
      const int xr_ = xrow0 + it * XSTEP;                                                                                     \
      if (xr_ >= xlo_ && xr_ < xhi_ && xr_ < XROWS) {                                                                         \
        const ptrdiff_t o_ = xbase_ + (ptrdiff_t)(it * XSTEP) * a.ldx;                                                        \
        if constexpr (XH) v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const dx_h16*>(a.X) + o_ + xtoff0);            \
        else v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.X) + o_ + xtoff0);                          \
      }                                                                                                                       \
      xreg[it] = v;                                                                                                           \
    }                                                                                                                         \
  }
#define DX_WG_STORE(IMG_OFF_)                                                                                                 \
  {                                                                                                                           \
    dx_h16* const dsw_ = Ds + (IMG_OFF_);                                                                                     \
    dx_h16* const xsw_ = Xs + (IMG_OFF_);                                                                                     \
    _Pragma("unroll") for (int it = 0; it < D_IT; ++it) {                                                                     \
      const int u = tid + it * NT;                                                                                           \
      const int row = u / DU, q = u % DU;                                                                                     \
      if constexpr (DYH) *reinterpret_cast<f32x4*>(dsw_ + row * WB_LD + ((q * 8) ^ WB_X(row))) = dreg[it];                    \
      else *reinterpret_cast<uint2*>(dsw_ + row * WB_LD + ((q * 4) ^ WB_X(row))) = pack_bf16x4v(dreg[it]);                    \
    }                                                                                                                         \
    _Pragma("unroll") for (int it = 0; it < X_IT; ++it) {                                                                     \
      const int u = tid + it * NT;                                                                                           \
      const int row = u / XU, q = u % XU;                                                                                     \
      if (u < XROWS * XU) {                                                                                                   \
        if constexpr (XH) *reinterpret_cast<f32x4*>(xsw_ + row * WB_LD + ((q * 8) ^ WB_X(row))) = xreg[it];                   \
        else *reinterpret_cast<uint2*>(xsw_ + row * WB_LD + ((q * 4) ^ WB_X(row))) = pack_bf16x4v(xreg[it]);                  \
      }                                                                                                                       \
    }                                                                                                                         \
  }

  // chunk walk of this split-K slice (interleaved: every slice sees a mix of utterance lengths).  The per-utterance limits sit in
  // LDS and (b, chunk-in-row) advance incrementally: a scalar global load + an integer division per chunk cost ~2 k cycles here.
  for (int i = tid; i < min(a.B, 512); i += NT) limit_s[i] = a.skip_halo >= 0 ? a.lens[i] + a.skip_halo : 0x7fffffff;
  __syncthreads();
  const int step_b = a.ksplit / chunks_per_row, step_k = a.ksplit - step_b * chunks_per_row;
  int c = split_id, b = c / chunks_per_row, kc = c - b * chunks_per_row, nc = 0;
  auto advance = [&]() { c += a.ksplit; b += step_b; kc += step_k; if (kc >= chunks_per_row) { kc -= chunks_per_row; ++b; } };
  auto live = [&]() {
    nc = kc * WB_BK;
    return nc < (b < 512 ? limit_s[b] : (a.skip_halo >= 0 ? a.lens[b] + a.skip_halo : 0x7fffffff));
  };
  while (c < total && !live()) advance();
  if constexpr (!DMA) if (c < total) DX_WG_LOAD(b, nc);
#ifndef DX_WG_ABL
#define DX_WG_ABL 0      // timing ablations (tools/ablation_build.py): 1 no MFMA, 2 no global loads in the loop, 4 tiles staged once
#endif
  if constexpr (DMA) {
    // ---- LDS-DMA ring -------------------------------------------------------------------------------------------------------------
    // Measured on the serial form (tools/microbench_wgrad.py, ablation builds, eight k = 3 jobs of the frame-level decoder): 277 us as it
    // stood, 218 without the global loads, 191 without the MFMAs, 103 with neither; this launch moves 509 MB for 177 GFLOP, i.e. it is
    // HBM-bound at its best (110 us at the 4.6 TB/s the load path reaches alone) and was running memory time + matrix time + epilogue.
    // A register-staged double-buffered pipeline did not fit (96 accumulators + 36 staging + 56 fragment registers spill, and a scratch
    // reload drains the whole load queue: stamped, tools/wgrad_stamps.py).  So the tiles never touch registers: every wave issues four
    // (wave 0: five) `global_load_lds_dwordx4` per 64-token chunk, each writing four 256-byte rows of an image; THREE chunks are in
    // flight while the fourth image is multiplied; counted `s_waitcnt vmcnt` + a raw barrier per chunk (a `__syncthreads()` would drain
    // the queue).  Rows that do not exist (utterance ends, halo) read a zero unit instead.  The image is unpadded, so the 16-byte units
    // of a row are XOR-swizzled by (row & 3) << 2 on the SOURCE side: the transposed fragment reads of four consecutive rows then hit
    // disjoint banks.  Fragments: K step ks + 1 is read before the MFMAs of step ks are issued (two register sets); the three taps of
    // an X fragment come from ONE 12-token window (three reads + four v_alignbit instead of six reads).
    constexpr int NKS = WB_BK / 32;                               // 2
    constexpr int XW = TAPS == 3 ? 3 : 2;
    static_assert(NKS == 2, "DMA ring: two K steps per chunk");
    // hipcc treats an LDS-DMA in flight as a pending write to ALL of LDS: any LDS read it can see gets an `s_waitcnt vmcnt(0)` in front
    // (seen in the ISA of a first version with the builtin transposed reads: the ring drained every chunk).  So inside the loop the
    // compiler sees NO LDS access: the fragment reads are inline asm with hand-placed lgkmcnt waits (the wait asm ties the fragment
    // registers, so no MFMA can be scheduled above it), the chunk walk reads lens[] through the scalar cache, and the bias gradient is
    // one more MFMA column (dY fragments x a vector of ones) instead of LDS column sums.
    struct Frags { s16x4 v[8 + 2 * XW]; };                        // dY (i, h) -> v[2 i + h]; X window (j, h) -> v[8 + XW j + h]
    Frags F[2];
    char* const smem_b = reinterpret_cast<char*>(smem_all);
    const unsigned smem_a = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem_b);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int li = lane & 15, fq = li >> 2, fp = li & 3;
    const int rowoff = (8 * g + fq) * 256 + (fp & 1) * 8;
    // Byte addresses inside image 0.  Swizzle: unit' = unit ^ ((row & 3) << 2) ^ (((row >> 3) & 1) << 1).  The first term spreads the four
    // rows of a 16-lane transposed read over disjoint banks; the second gives the two 16-lane groups of a half-wave (rows 8 apart: the
    // same banks in a 256-byte-row image) complementary halves of the bank row - without it the fragment reads ran at ~80 B/clk (2-way
    // conflict, measured: the reads alone took as long as the MFMAs).  Fragment rows are 32 ks + 8 g + q + 4 h: (row >> 3) & 1 = g & 1,
    // except for the third window read (h = 2), which belongs to the next group of eight.
    const int gb = g & 1;
    unsigned dA[4], xA[2], xA2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) dA[i] = smem_a + rowoff + (((wc * 8 + i * 2 + (fp >> 1)) ^ (fq << 2) ^ (gb << 1)) << 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      xA[j] = smem_a + WB_BK * 256 + rowoff + (((wt * 4 + j * 2 + (fp >> 1)) ^ (fq << 2) ^ (gb << 1)) << 4);
      xA2[j] = smem_a + WB_BK * 256 + rowoff + (((wt * 4 + j * 2 + (fp >> 1)) ^ (fq << 2) ^ ((gb ^ 1) << 1)) << 4);
    }
#define DX_RD(F_, SLOT_, ADR_, OFF_) F_.v[SLOT_] = dx_tr16_b64_asm<(OFF_)>(ADR_);
#define DX_READ_FRAGS(F_, IMG_, KS_)                                                                                          \
  {                                                                                                                           \
    const unsigned o_ = (unsigned)((IMG_) * DMA_IMGB);                                                                        \
    DX_RD(F_, 0, dA[0] + o_, ((KS_) * 32) * 256) DX_RD(F_, 1, dA[0] + o_, ((KS_) * 32 + 4) * 256)                             \
    DX_RD(F_, 2, dA[1] + o_, ((KS_) * 32) * 256) DX_RD(F_, 3, dA[1] + o_, ((KS_) * 32 + 4) * 256)                             \
    DX_RD(F_, 4, dA[2] + o_, ((KS_) * 32) * 256) DX_RD(F_, 5, dA[2] + o_, ((KS_) * 32 + 4) * 256)                             \
    DX_RD(F_, 6, dA[3] + o_, ((KS_) * 32) * 256) DX_RD(F_, 7, dA[3] + o_, ((KS_) * 32 + 4) * 256)                             \
    DX_RD(F_, 8, xA[0] + o_, ((KS_) * 32) * 256) DX_RD(F_, 9, xA[0] + o_, ((KS_) * 32 + 4) * 256)                             \
    if constexpr (XW == 3) DX_RD(F_, 10, xA2[0] + o_, ((KS_) * 32 + 8) * 256)                                                 \
    DX_RD(F_, 8 + XW, xA[1] + o_, ((KS_) * 32) * 256) DX_RD(F_, 9 + XW, xA[1] + o_, ((KS_) * 32 + 4) * 256)                   \
    if constexpr (XW == 3) DX_RD(F_, 10 + XW, xA2[1] + o_, ((KS_) * 32 + 8) * 256)                                            \
  }
    // every fragment of the set has landed; ties the registers so that their consumers stay below
    auto landed = [&](Frags& f) {
      if constexpr (XW == 3)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.v[0]), "+v"(f.v[1]), "+v"(f.v[2]), "+v"(f.v[3]), "+v"(f.v[4]), "+v"(f.v[5]), "+v"(f.v[6]), "+v"(f.v[7]),
                     "+v"(f.v[8]), "+v"(f.v[9]), "+v"(f.v[10]), "+v"(f.v[11]), "+v"(f.v[12]), "+v"(f.v[13]) : : "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.v[0]), "+v"(f.v[1]), "+v"(f.v[2]), "+v"(f.v[3]), "+v"(f.v[4]), "+v"(f.v[5]), "+v"(f.v[6]), "+v"(f.v[7]),
                     "+v"(f.v[8]), "+v"(f.v[9]), "+v"(f.v[10]), "+v"(f.v[11]) : : "memory");
    };
    f32x4 accb[4];                                                // bias gradient: dY fragment x ones, the waves of input-channel slice 0
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool bias_wave = do_bias && __builtin_amdgcn_readfirstlane(wt) == 0;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (dx_h16)1.0f;
    auto multiply = [&](const Frags& f) {
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      typedef int i32x2 __attribute__((ext_vector_type(2)));
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      bf16x8 df[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const s16x4 lo = f.v[2 * i], hi = f.v[2 * i + 1];
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        df[i] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        bf16x8 xf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // window dwords w0..w4 = tokens (0,1) (2,3) (4,5) (6,7) (8,9) of this lane's channel
          const i32x2 a0 = __builtin_bit_cast(i32x2, f.v[8 + XW * j]), a1 = __builtin_bit_cast(i32x2, f.v[9 + XW * j]);
          int w[5] = {a0[0], a0[1], a1[0], a1[1], 0};
          if constexpr (TAPS == 3) w[4] = __builtin_bit_cast(i32x2, f.v[8 + XW * j + XW - 1])[0];
          i32x4 o;
          if (t == 0) o = i32x4{w[0], w[1], w[2], w[3]};
          else if (t == 2) o = i32x4{w[1], w[2], w[3], w[4]};
          else o = i32x4{(int)__builtin_amdgcn_alignbit((unsigned)w[1], (unsigned)w[0], 16), (int)__builtin_amdgcn_alignbit((unsigned)w[2], (unsigned)w[1], 16),
                         (int)__builtin_amdgcn_alignbit((unsigned)w[3], (unsigned)w[2], 16), (int)__builtin_amdgcn_alignbit((unsigned)w[4], (unsigned)w[3], 16)};
          xf[j] = __builtin_bit_cast(bf16x8, o);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (DX_WG_ABL & 1) asm volatile("" :: "v"(df[i]), "v"(xf[j]));
            else acc[t][i][j] = DX_MFMA_H16(df[i], xf[j], acc[t][i][j]);
          }
      }
      if (bias_wave) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = DX_MFMA_H16(df[i], ones, accb[i]);
      }
    };
    // DMA source addressing: lane -> (row in the group of four, unit of the row); the unit index is swizzled on the source side
    const int srow = lane >> 4, sunit = (lane & 15) ^ (srow << 2) ^ (((wave_u >> 1) & 1) << 1);     // this wave's groups: wave + 8 k, rows 4 * group + srow
    const bool dch_ok = co0 + sunit * 8 < a.Cout, xch_ok = ci0 + sunit * 8 < a.Cin;
    const int dsoff = srow * a.ldy + co0 + sunit * 8, xsoff = srow * a.ldx + ci0 + sunit * 8;     // elements
    const dx_h16* const zero_src = reinterpret_cast<const dx_h16*>(dx_wgrad_zero_unit);
    auto issue = [&](int img, int bb, int nn) {
      const dx_h16* const dsrc = reinterpret_cast<const dx_h16*>(a.dY) + ((ptrdiff_t)bb * a.N + nn) * a.ldy;
      const dx_h16* const xsrc = reinterpret_cast<const dx_h16*>(a.X) + ((ptrdiff_t)bb * a.N + nn - PAD) * a.ldx;
      const int dlim = a.N - nn, xlo = PAD - nn, xhi = min(a.N - nn + PAD, WB_BK + TAPS - 1);
      char* const ibase = smem_b + img * DMA_IMGB;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int grp = wave_u + 8 * k, row = 4 * grp + srow;
        const dx_h16* src = (dch_ok && row < dlim) ? dsrc + (ptrdiff_t)(4 * grp) * a.ldy + dsoff : zero_src;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(ibase + grp * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int grp = wave_u + 8 * k;
        if (grp < DMA_XG) {                                       // wave-uniform: k = 2 is wave 0's (k = 3 layers)
          const int row = 4 * grp + srow;
          const dx_h16* src = (xch_ok && row >= xlo && row < xhi) ? xsrc + (ptrdiff_t)(4 * grp) * a.ldx + xsoff : zero_src;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(ibase + WB_BK * 256 + grp * 1024), 16, 0, 0);
        }
      }
    };
    // this wave's DMA instructions per chunk: 4, or 5 for wave 0 of a k = 3 layer; `ahead` chunks may stay in flight
    auto wait_dma = [&](int ahead) {
      if (DMA_XG > 16 && wave_u == 0) {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    // the chunk walk of this form: limits through the scalar cache (no LDS read the compiler can see, see above)
    auto live_s = [&]() {
      b = __builtin_amdgcn_readfirstlane(b); kc = __builtin_amdgcn_readfirstlane(kc); c = __builtin_amdgcn_readfirstlane(c);   // provably uniform: s_load, scalar branch
      nc = kc * WB_BK;
      if (a.skip_halo < 0) return true;
      int len;                               // hipcc will not use the scalar cache by itself here (the kernel's atomics may alias lens[])
      asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(len) : "s"(a.lens + b) : "memory");
      return nc < len + a.skip_halo;
    };
    while (c < total && !live_s()) advance();
    { WG_STAMP(0, t_start) WG_STAMP(1, WG_NOW()) }
#ifdef DX_WG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stamp stores must not sit in the counted DMA queue
#endif
    int issued = 0, done = 0;
#pragma unroll 1
    for (int k = 0; k < DMA_NIMG - 1 && c < total; ++k) {
      issue(k, b, nc);
      ++issued;
      advance();
      while (c < total && !live_s()) advance();
    }
    if (issued) {
      wait_dma(issued - 1);
      __builtin_amdgcn_s_barrier();
      DX_READ_FRAGS(F[0], 0, 0)
    }
#pragma unroll 1
    while (done < issued) {
      const int img = done & (DMA_NIMG - 1);
      if (c < total) {                       // image (done + 3) & 3 was last read before the barrier of the previous iteration
        if (!(DX_WG_ABL & 2)) issue((done + DMA_NIMG - 1) & (DMA_NIMG - 1), b, nc);
        ++issued;
        advance();
        while (c < total && !live_s()) advance();
      }
      landed(F[0]);                          // read a whole multiply ago (and drains the walk's scalar loads)
      DX_READ_FRAGS(F[1], img, 1)
      multiply(F[0]);
      const bool more = done + 1 < issued;
#ifdef DX_WG_STAMPS
      const unsigned long long tv0 = WG_NOW();
#endif
      if (more && !(DX_WG_ABL & 2)) wait_dma(issued - done - 2);
#ifdef DX_WG_STAMPS
      const unsigned long long tb0 = WG_NOW();
      t_vm += tb0 - tv0;
#endif
      landed(F[1]);
      __builtin_amdgcn_s_barrier();          // chunk done + 1 is complete in LDS; nobody reads image img any more
#ifdef DX_WG_STAMPS
      t_bar += WG_NOW() - tb0; ++n_chunks;
#endif
      if (more) DX_READ_FRAGS(F[0], (done + 1) & (DMA_NIMG - 1), 0)
      multiply(F[1]);
      ++done;
    }
#undef DX_RD
#undef DX_READ_FRAGS
    if (bias_wave) {                         // lanes of output column 0 hold the sums of rows 4 g + e
      if (r == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int co = co0 + wc * 64 + i * 16 + g * 4 + e;
            if (co < a.Cout && accb[i][e] != 0.f) atomicAdd(&a.dbias[co], accb[i][e]);
          }
      }
    }
    __syncthreads();                         // every DMA was waited for above: the epilogue may reuse the images
    WG_STAMP(2, WG_NOW()) WG_STAMP(4, t_vm) WG_STAMP(5, t_bar) WG_STAMP(6, n_chunks)
  } else if constexpr (PIPE) {
    // Software pipeline over two LDS images, ONE barrier per chunk.  The serial form below (stage, barrier, read fragments, multiply,
    // barrier) measured, on the eight k = 3 jobs of the frame-level decoder: 279 us as it stands, 185 without the MFMAs, 175 without the
    // global loads and the staging, 100 with neither - i.e. the three parts ran one AFTER the other (ablation builds, tools/microbench_wgrad.py).
    // Here, while chunk n is multiplied out of image p: the fragments of K step ks + 1 are read before the MFMAs of step ks are issued
    // (two fragment sets in registers); chunk n + 1 (in the staging registers since the previous iteration) is stored into image p ^ 1
    // after the first K step and the global loads of chunk n + 2 are issued behind it (a full chunk of cover); the barrier sits before
    // the LAST K step's MFMAs, so the first fragments of chunk n + 1 are read under them.
    // LDS reads: the three taps of an X fragment are the SAME 8 + 2 tokens of a channel shifted by one - one 12-token window (three
    // transposed reads) and four v_alignbit per fragment instead of six reads: 14 instead of 20 reads per 24 MFMAs (the LDS read
    // pipe, 128 B/clk for these reads, is as long as the MFMA time in the serial form).
    constexpr int NKS = WB_BK / 32;
    constexpr int XW = TAPS == 3 ? 3 : 2;                         // transposed 4-token reads per X fragment window
    struct Frags { s16x4 d[4][2]; s16x4 x[2][XW]; };
    Frags F[2];
    const int li = lane & 15, fq = li >> 2, fp = li & 3;
    const int lane_off = (8 * g + fq) * WB_LD + 4 * fp;
    const int gx = (g & 1) << 6;                                  // WB_X of this lane's rows 32 ks + 8 g + fq + {0, 4}; the third window read (+ 8) is in the next group of eight
    const dx_h16* const dbase = Ds + lane_off + ((wc * 64) ^ gx);
    const dx_h16* const xbase = Xs + lane_off + ((wt * 32) ^ gx);
    const dx_h16* const xbase2 = Xs + lane_off + ((wt * 32) ^ gx ^ 64);
    auto tr4 = [](const dx_h16* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p)); };
    auto read_frags = [&](int img_off, int ks, Frags& f) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) f.d[i][h] = tr4(dbase + img_off + (ks * 32 + 4 * h) * WB_LD + i * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int h = 0; h < XW; ++h) f.x[j][h] = tr4((h == 2 ? xbase2 : xbase) + img_off + (ks * 32 + 4 * h) * WB_LD + j * 16);
    };
    auto multiply = [&](const Frags& f) {
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      typedef int i32x2 __attribute__((ext_vector_type(2)));
      bf16x8 df[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const s16x8 v = {f.d[i][0][0], f.d[i][0][1], f.d[i][0][2], f.d[i][0][3], f.d[i][1][0], f.d[i][1][1], f.d[i][1][2], f.d[i][1][3]};
        df[i] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        bf16x8 xf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // window dwords w0..w5 = tokens (0,1) (2,3) (4,5) (6,7) (8,9) (10,11) of this lane's channel
          const i32x2 a0 = __builtin_bit_cast(i32x2, f.x[j][0]), a1 = __builtin_bit_cast(i32x2, f.x[j][1]);
          int w[5] = {a0[0], a0[1], a1[0], a1[1], 0};
          if constexpr (TAPS == 3) w[4] = __builtin_bit_cast(i32x2, f.x[j][2])[0];
          typedef int i32x4 __attribute__((ext_vector_type(4)));
          i32x4 o;
          if (t == 0) o = i32x4{w[0], w[1], w[2], w[3]};
          else if (t == 2) o = i32x4{w[1], w[2], w[3], w[4]};
          else o = i32x4{(int)__builtin_amdgcn_alignbit((unsigned)w[1], (unsigned)w[0], 16), (int)__builtin_amdgcn_alignbit((unsigned)w[2], (unsigned)w[1], 16),
                         (int)__builtin_amdgcn_alignbit((unsigned)w[3], (unsigned)w[2], 16), (int)__builtin_amdgcn_alignbit((unsigned)w[4], (unsigned)w[3], 16)};
          xf[j] = __builtin_bit_cast(bf16x8, o);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (DX_WG_ABL & 1) asm volatile("" :: "v"(df[i]), "v"(xf[j]));
            else acc[t][i][j] = DX_MFMA_H16(df[i], xf[j], acc[t][i][j]);
          }
      }
    };
    bool have = c < total;                 // a chunk is in the staging registers
    if (have) {
      DX_WG_STORE(0);
      advance();
      while (c < total && !live()) advance();
      if (c < total) DX_WG_LOAD(b, nc);
      __syncthreads();
      read_frags(0, 0, F[0]);
    }
    int img = 0;                           // element offset of the image being multiplied
    { [[maybe_unused]] const int tid = threadIdx.x; WG_STAMP(0, t_start) WG_STAMP(1, WG_NOW()) }
    while (have) {
      const bool next = c < total;         // the staging registers hold (or wait for) the chunk after this one
      if (do_bias) {
        const int r0b = (tid >> 6) * (WB_BK / (NT / 64));
        const dx_h16* col = Ds + img + r0b * WB_LD;
#pragma unroll
        for (int k = 0; k < WB_BK / (NT / 64); ++k) {
          const unsigned v = *reinterpret_cast<const unsigned*>(col + k * WB_LD + (((tid & 63) * 2) ^ WB_X(r0b + k)));
          bsum0 += (float)__builtin_bit_cast(dx_h16, (unsigned short)(v & 0xffffu));
          bsum1 += (float)__builtin_bit_cast(dx_h16, (unsigned short)(v >> 16));
        }
      }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        if (ks + 1 < NKS) read_frags(img, ks + 1, F[(ks + 1) & 1]);
        else {
#ifdef DX_WG_STAMPS
          const unsigned long long tb0 = WG_NOW();
#endif
          __syncthreads();                 // image img ^ 1 is complete; nobody reads image img any more (the last fragments are in registers)
#ifdef DX_WG_STAMPS
          t_bar += WG_NOW() - tb0; ++n_chunks;
#endif
          if (next) read_frags(IMG - img, 0, F[0]);
        }
        if (ks == (NKS == 4 ? 1 : 0) && next) {
#ifdef DX_WG_STAMPS
          { const unsigned long long tv0 = WG_NOW(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_vm += WG_NOW() - tv0; }
#endif
          if (!(DX_WG_ABL & 4)) DX_WG_STORE(IMG - img);
          advance();
          while (c < total && !live()) advance();
          if (!(DX_WG_ABL & 2)) if (c < total) DX_WG_LOAD(b, nc);
        }
        multiply(F[ks & 1]);
      }
      have = next;
      img = IMG - img;
    }
    WG_STAMP(2, WG_NOW()) WG_STAMP(4, t_vm) WG_STAMP(5, t_bar) WG_STAMP(6, n_chunks)
  } else {
  [[maybe_unused]] bool first_chunk = true;
  while (c < total) {
    if (!(DX_WG_ABL & 4) || first_chunk) {
      __syncthreads();
      DX_WG_STORE(0);
      __syncthreads();
    }
    first_chunk = false;
    advance();
    while (c < total && !live()) advance();
    if (!(DX_WG_ABL & 2)) if (c < total) DX_WG_LOAD(b, nc);
    if (do_bias) {
      const int r0b = (tid >> 6) * (WB_BK / (NT / 64));
      const dx_h16* col = Ds + r0b * WB_LD;
#pragma unroll
      for (int k = 0; k < WB_BK / (NT / 64); ++k) {
        const unsigned v = *reinterpret_cast<const unsigned*>(col + k * WB_LD + (((tid & 63) * 2) ^ WB_X(r0b + k)));
        bsum0 += (float)__builtin_bit_cast(dx_h16, (unsigned short)(v & 0xffffu));
        bsum1 += (float)__builtin_bit_cast(dx_h16, (unsigned short)(v >> 16));
      }
    }
#pragma unroll
    for (int ks = 0; ks < WB_BK / 32; ++ks) {
      bf16x8 df[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) df[i] = tr_fragment(Ds, ks * 32, wc * 64 + i * 16, lane);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        bf16x8 xf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) xf[j] = tr_fragment(Xs, ks * 32 + t, wt * 32 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (DX_WG_ABL & 1) asm volatile("" :: "v"(df[i]), "v"(xf[j]));
            else acc[t][i][j] = DX_MFMA_H16(df[i], xf[j], acc[t][i][j]);
          }
      }
    }
  }
  }
#undef DX_WG_LOAD
#undef DX_WG_STORE
  // Epilogue: the partial sums go into the gradient in the PARAMETER's own layout (Cout, Cin, taps) - no re-layout launch, no
  // scratch tensor.  A lane holds 4 output channels x 1 input channel per accumulator, i.e. its addresses in that layout are
  // 4 rows x 12-byte strides; instead the tile is transposed through LDS, 32 (or 16) output-channel rows at a time, and every
  // wave-instruction adds 64 CONSECUTIVE floats of one row (256 contiguous bytes: the full-rate shape of the memory-side float
  // atomics, MI355X_MICROARCH.md "Global float atomics").
  if constexpr (TAPS == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + wt * 32 + j * 16 + r;
        if (ci >= a.Cin) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = co0 + wc * 64 + i * 16 + g * 4 + e;
          if (co < a.Cout && acc[0][i][j][e] != 0.f) atomicAdd(&a.G[(size_t)co * a.Cin + ci], acc[0][i][j][e]);
        }
      }
  } else {
    constexpr int ROWF = CI_T * TAPS;                              // floats per staged row: [ci][tap]
    constexpr int RPP = (SMEM_BYTES >= 32 * ROWF * 4) ? 32 : 16;   // rows per pass
    float* const stage = reinterpret_cast<float*>(smem_all);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int half = 0; half < 32 / RPP; ++half) {
        __syncthreads();                                           // chunk loop / previous pass: every LDS read has retired
        if (RPP == 32 || wc == half) {
          const int rowbase = RPP == 32 ? wc * 16 : 0;
#pragma unroll
          for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                stage[(rowbase + g * 4 + e) * ROWF + (wt * 32 + j * 16 + r) * TAPS + t] = acc[t][i][j][e];
        }
        __syncthreads();
        for (int u = tid; u < RPP * ROWF; u += NT) {               // ROWF % 64 == 0: a wave-instruction stays inside one row
          const int row = u / ROWF, c = u - row * ROWF;
          const int co = co0 + (RPP == 32 ? (row >> 4) * 64 + i * 16 + (row & 15) : half * 64 + i * 16 + row);
          const float v = stage[u];
          if (co < a.Cout && ci0 + c / TAPS < a.Cin && v != 0.f) atomicAdd(&a.G[((size_t)co * a.Cin + ci0) * TAPS + c], v);
        }
      }
    }
  }
  if (do_bias && !DMA) {                                 // workgroup-uniform: fold the row slices, one atomic per channel
    float* red = reinterpret_cast<float*>(Ds);           // the chunk loop and the staging passes are over
    __syncthreads();
    red[(tid >> 6) * TILE + (tid & 63) * 2] = bsum0;
    red[(tid >> 6) * TILE + (tid & 63) * 2 + 1] = bsum1;
    __syncthreads();
    if (tid < TILE && co0 + tid < a.Cout) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < NT / 64; ++q) t += red[q * TILE + tid];
      if (t != 0.f) atomicAdd(&a.dbias[co0 + tid], t);
    }
  }
#ifdef DX_WG_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  WG_STAMP(3, WG_NOW())
#endif
}

// ------------------------------------------------------------------------------------------------
// packing: checkpoint layout (Cout, Cin, TAPS) fp32 -> fwd [TAPS][CoutP][CinP], bwd [TAPS][CinPo][CoutPi]
// (bwd = transposed channels, flipped taps: the input-gradient conv).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ W, T* __restrict__ fwd, T* __restrict__ bwd,
                                    int Cout, int Cin, int taps, int CoutP_f, int CinP_f, int CinP_b, int CoutP_b) {
  const size_t nf = (size_t)taps * CoutP_f * CinP_f;
  const size_t nb = bwd ? (size_t)taps * CinP_b * CoutP_b : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += (size_t)gridDim.x * blockDim.x) {
    if (i < nf) {
      const int ci = (int)(i % CinP_f);
      const int co = (int)((i / CinP_f) % CoutP_f);
      const int tap = (int)(i / ((size_t)CinP_f * CoutP_f));
      const float v = (co < Cout && ci < Cin) ? W[((size_t)co * Cin + ci) * taps + tap] : 0.f;
      fwd[sizeof(T) == 2 ? wb_off(tap, co, ci, CoutP_f, CinP_f) : i] = (T)v;
    } else {
      const size_t k = i - nf;
      const int co = (int)(k % CoutP_b);
      const int ci = (int)((k / CoutP_b) % CinP_b);
      const int tap = (int)(k / ((size_t)CoutP_b * CinP_b));
      const float v = (co < Cout && ci < Cin) ? W[((size_t)co * Cin + ci) * taps + (taps - 1 - tap)] : 0.f;
      bwd[sizeof(T) == 2 ? wb_off(tap, ci, co, CinP_b, CoutP_b) : k] = (T)v;
    }
  }
}

// all layers of a model in ONE launch: blockIdx.y = layer, descriptors in device memory
struct PackDesc {
  const float* W; void* fwd; void* bwd;
  int Cout, Cin, taps, CoutP_f, CinP_f, CinP_b, CoutP_b, pad;
};
template <typename T>
__global__ void pack_weights_batched_kernel(const PackDesc* __restrict__ descs) {
  const PackDesc d = descs[blockIdx.y];
  T* fwd = reinterpret_cast<T*>(d.fwd);
  T* bwd = reinterpret_cast<T*>(d.bwd);
  const size_t nf = (size_t)d.taps * d.CoutP_f * d.CinP_f;
  const size_t nb = bwd ? (size_t)d.taps * d.CinP_b * d.CoutP_b : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += (size_t)gridDim.x * blockDim.x) {
    if (i < nf) {
      const int ci = (int)(i % d.CinP_f);
      const int co = (int)((i / d.CinP_f) % d.CoutP_f);
      const int tap = (int)(i / ((size_t)d.CinP_f * d.CoutP_f));
      fwd[sizeof(T) == 2 ? wb_off(tap, co, ci, d.CoutP_f, d.CinP_f) : i] = (T)((co < d.Cout && ci < d.Cin) ? d.W[((size_t)co * d.Cin + ci) * d.taps + tap] : 0.f);
    } else {
      const size_t k = i - nf;
      const int co = (int)(k % d.CoutP_b);
      const int ci = (int)((k / d.CoutP_b) % d.CinP_b);
      const int tap = (int)(k / ((size_t)d.CoutP_b * d.CinP_b));
      bwd[sizeof(T) == 2 ? wb_off(tap, ci, co, d.CinP_b, d.CoutP_b) : k] = (T)((co < d.Cout && ci < d.Cin) ? d.W[((size_t)co * d.Cin + ci) * d.taps + (d.taps - 1 - tap)] : 0.f);
    }
  }
}

// bf16 form of the batched pack: one wave builds one 16-row x 32-k block of the fragment-major pack for every tap, so each
// store is a contiguous 1 KB wave-write (the element-order loop above wrote 16-byte pieces 256 B apart and walked the
// checkpoint with a 12-byte stride once per tap: 105 us per step for 57 MB of weights).  Lane = (row r, k-group g): forward
// blocks read 8 consecutive input channels x taps (96 B contiguous per lane, 4 lanes per row), backward blocks (rows = input
// channels, k = output channels, taps flipped) read 8 output-channel rows at one input channel.
// one 16-row x 32-k block (all taps) of descriptor d: blk < nfb forward, else backward
__device__ __forceinline__ void pack_block_bf16(const PackDesc& d, int blk, int lane) {
  dx_h16* fwd = reinterpret_cast<dx_h16*>(d.fwd);
  dx_h16* bwd = reinterpret_cast<dx_h16*>(d.bwd);
  const int r = lane & 15, g = lane >> 4;
  const int fkb = d.CinP_f >> 5, nfb = (d.CoutP_f >> 4) * fkb;          // forward blocks per tap
  const int bkb = d.CoutP_b >> 5, nbb = bwd ? (d.CinP_b >> 4) * bkb : 0;
  if (blk < nfb) {
    const int co = (blk / fkb) * 16 + r, ci = (blk % fkb) * 32 + g * 8;
    // all 24 loads are unconditional (indices clamped, zeros selected in afterwards): a load inside a lane-dependent branch makes the
    // compiler wait for each one before the next (24 round trips per wave: this kernel took 68 us for 115 MB)
    float v[3][8];
    const int tl = d.taps - 1;
    const size_t rowb = (size_t)min(co, d.Cout - 1) * d.Cin;
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 3; ++t) v[t][e] = d.W[(rowb + min(ci + e, d.Cin - 1)) * d.taps + min(t, tl)];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t >= d.taps) break;
      bf16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (dx_h16)((co < d.Cout && ci + e < d.Cin) ? v[t][e] : 0.f);
      *reinterpret_cast<bf16x8*>(fwd + ((size_t)t * nfb + blk) * 512 + lane * 8) = h;
    }
  } else if (blk < nfb + nbb) {
    const int bb = blk - nfb;
    const int ci = (bb / bkb) * 16 + r, co = (bb % bkb) * 32 + g * 8;
    float v[3][8];
    const int tl = d.taps - 1, cic = min(ci, d.Cin - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 3; ++t) v[t][e] = d.W[((size_t)min(co + e, d.Cout - 1) * d.Cin + cic) * d.taps + min(t, tl)];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t >= d.taps) break;
      bf16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (dx_h16)((ci < d.Cin && co + e < d.Cout) ? (d.taps == 3 ? v[2 - t][e] : v[0][e]) : 0.f);    // taps flipped
      *reinterpret_cast<bf16x8*>(bwd + ((size_t)t * nbb + bb) * 512 + lane * 8) = h;
    }
  }
}

__global__ __launch_bounds__(256) void pack_weights_batched_bf16_kernel(const PackDesc* __restrict__ descs) {
  const PackDesc d = descs[blockIdx.y];
  const int fkb = d.CinP_f >> 5, nfb = (d.CoutP_f >> 4) * fkb;
  const int bkb = d.CoutP_b >> 5, nbb = d.bwd ? (d.CinP_b >> 4) * bkb : 0;
  for (int blk = blockIdx.x * 4 + (threadIdx.x >> 6); blk < nfb + nbb; blk += gridDim.x * 4) pack_block_bf16(d, blk, threadIdx.x & 63);
}

// The same work with the descriptors as KERNEL ARGUMENTS and a flat grid: one wave per block of the whole model.  The form above
// launches 512 workgroups per layer whatever its size (layers differ by four orders of magnitude) and every workgroup starts with a
// global round trip for its descriptor: 29 k workgroups, most of which fetch a descriptor and leave - 71 us per step for 115 MB.
constexpr int PACK_MAX = 60;          // 60 x 56 B + prefix: 3.6 KB of the 4 KB argument segment (the runtime's hidden arguments take 256 B)
struct PackBatchArgs {
  PackDesc d[PACK_MAX];
  int prefix[PACK_MAX + 1];          // exclusive prefix of blocks per descriptor; prefix[n] = total
  int n;
};
static_assert(sizeof(PackBatchArgs) <= 3712, "kernel arguments must stay under the 4 KB limit");
__global__ __launch_bounds__(256) void pack_weights_flat_bf16_kernel(const PackBatchArgs a) {
  const int wb = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: scalar loads from the argument block
  if (wb >= a.prefix[a.n]) return;
  int lo = 0, hi = a.n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.prefix[mid] <= wb) lo = mid; else hi = mid;
  }
  pack_block_bf16(a.d[lo], wb - a.prefix[lo], threadIdx.x & 63);
}

// grad[co][ci][tap] (+)= G[tap][co][ci]
__global__ void unpack_wgrad_kernel(const float* __restrict__ G, float* __restrict__ grad, int Cout, int Cin, int taps, int accumulate) {
  const size_t n = (size_t)Cout * Cin * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const size_t cc = i / taps;
    const float v = G[(size_t)tap * Cout * Cin + cc];
    grad[i] = accumulate ? grad[i] + v : v;
  }
}

// column sums: out[c] += sum_rows X[row][c]   (bias gradients); X fp32 or bf16
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int ldx, float* __restrict__ out,
                                                     long rows, int C, int rows_per_block) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = min(rows, r0 + rows_per_block);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long rr = r0;
  for (; rr + 3 < r1; rr += 4) {
    s0 += (float)X[(size_t)rr * ldx + c];
    s1 += (float)X[(size_t)(rr + 1) * ldx + c];
    s2 += (float)X[(size_t)(rr + 2) * ldx + c];
    s3 += (float)X[(size_t)(rr + 3) * ldx + c];
  }
  for (; rr < r1; ++rr) s0 += (float)X[(size_t)rr * ldx + c];
  const float t = (s0 + s1) + (s2 + s3);
  if (t != 0.f) atomicAdd(&out[c], t);
}

template <typename T, int TAPS, int TOK, bool XH>
void launch_conv_inst(const ConvGemmArgs& a, hipStream_t s) {
  const size_t smem = (size_t)(TAPS * TILE + TOK + TAPS - 1) * 128;
  static bool configured = false;
  if (!configured) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, TAPS, TOK, XH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  dim3 grid(a.B * dx_cdiv(a.N, TOK), a.CoutP / TILE);
  hipLaunchKernelGGL((conv_gemm_kernel<T, TAPS, TOK, XH>), grid, dim3(256), smem, s, a);
}

template <typename T, int TAPS>
int launch_conv(const ConvGemmArgs& a, hipStream_t s) {
  // narrow outputs (Cout <= 128: conv2, out-proj, mel projection ...) launch few workgroups: in exact-f32 mode (MFMA-paced)
  // 64-token tiles fill the chip better (measured 405 -> 354 us on the FF conv2); in bf16 mode the 128-token tile stays ahead
  // (62 vs 79 us) because the weight tile is re-staged half as often.
  // (bf16, k = 1, few workgroups - the phoneme-level Linear layers: 64-token tiles measured 9.6 -> 7.7 us; bf16 k = 3 stays at 128)
  static const long small_below = getenv("DX_CONV_SMALL_BELOW") ? atol(getenv("DX_CONV_SMALL_BELOW")) : ((sizeof(T) == 4 || TAPS == 1) ? 1024 : 0);
  const bool small = (long)a.B * dx_cdiv(a.N, TILE) * (a.CoutP / TILE) < small_below;
  if constexpr (sizeof(T) == 2) {
    if (a.x_bf16) { if (small) launch_conv_inst<T, TAPS, 64, true>(a, s); else launch_conv_inst<T, TAPS, 128, true>(a, s); return DX_OK; }
  }
  if (small) launch_conv_inst<T, TAPS, 64, false>(a, s); else launch_conv_inst<T, TAPS, 128, false>(a, s);
  return DX_OK;
}

}  // namespace

extern "C" {

// Packed-weight geometry for a (Cout, Cin, taps) layer at operand precision `bf16` (0: f32, 1: bf16).
// Returns element counts so the caller can allocate; dims: [CoutP_f, CinP_f, CinP_b, CoutP_b].
int dx_pack_dims(int Cout, int Cin, int bf16, int* dims) {
  const int bk = bf16 ? 64 : 32;
  dims[0] = dx_roundup(Cout, TILE);
  dims[1] = dx_roundup(Cin, bk);
  dims[2] = dx_roundup(Cin, TILE);
  dims[3] = dx_roundup(Cout, bk);
  return DX_OK;
}

int dx_pack_weights(const float* W, void* fwd, void* bwd, int Cout, int Cin, int taps, int bf16, void* stream) {
  DX_REQUIRE(W && fwd, "dx_pack_weights: null pointer");
  DX_REQUIRE(Cout > 0 && Cin > 0 && (taps == 1 || taps == 3), "dx_pack_weights: bad dims Cout=%d Cin=%d taps=%d", Cout, Cin, taps);
  int d[4];
  dx_pack_dims(Cout, Cin, bf16, d);
  const size_t n = (size_t)taps * d[0] * d[1] + (bwd ? (size_t)taps * d[2] * d[3] : 0);
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipStream_t s = (hipStream_t)stream;
  if (bf16)
    hipLaunchKernelGGL(pack_weights_kernel<dx_h16>, dim3(blocks), dim3(256), 0, s, W, (dx_h16*)fwd, (dx_h16*)bwd, Cout, Cin, taps, d[0], d[1], d[2], d[3]);
  else
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, s, W, (float*)fwd, (float*)bwd, Cout, Cin, taps, d[0], d[1], d[2], d[3]);
  DX_LAUNCH_CHECK("dx_pack_weights");
  return DX_OK;
}

// descs: device array of n records {W, fwd, bwd (8-byte pointers), Cout, Cin, taps, CoutP_f, CinP_f, CinP_b, CoutP_b, pad (int32)}
int dx_pack_weights_batched(const void* descs, int n, int bf16, void* stream) {
  DX_REQUIRE(descs && n > 0, "dx_pack_weights_batched: bad arguments");
  static_assert(sizeof(PackDesc) == 56, "PackDesc layout is part of the ABI");
  dim3 grid(512, n);   // layers differ by 4 orders of magnitude in size: surplus blocks of the small ones exit at once
  if (bf16) hipLaunchKernelGGL(pack_weights_batched_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs);
  else hipLaunchKernelGGL(pack_weights_batched_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs);
  DX_LAUNCH_CHECK("dx_pack_weights_batched");
  return DX_OK;
}

// descs: HOST array of n 56-byte records (as above).  16-bit packs: the descriptors travel as kernel arguments (no device table) and the
// grid holds one wave per 16 x 32 block of the whole model; exact-f32 packs: the records are staged through a small device table.
int dx_pack_weights_host(const void* descs, int n, int bf16, void* stream) {
  DX_REQUIRE(descs && n > 0, "dx_pack_weights_host: bad arguments");
  const PackDesc* h = reinterpret_cast<const PackDesc*>(descs);
  hipStream_t s = (hipStream_t)stream;
  if (!bf16) {
    for (int i = 0; i < n; ++i)
      if (int rc = dx_pack_weights(h[i].W, h[i].fwd, h[i].bwd, h[i].Cout, h[i].Cin, h[i].taps, 0, stream)) return rc;
    return DX_OK;
  }
  for (int base = 0; base < n; base += PACK_MAX) {
    PackBatchArgs a{};
    a.n = std::min(PACK_MAX, n - base);
    int run = 0;
    for (int i = 0; i < a.n; ++i) {
      const PackDesc& d = h[base + i];
      DX_REQUIRE(d.W && d.fwd && d.Cout > 0 && d.Cin > 0 && (d.taps == 1 || d.taps == 3), "dx_pack_weights_host: bad descriptor %d", base + i);
      a.d[i] = d;
      a.prefix[i] = run;
      run += (d.CoutP_f >> 4) * (d.CinP_f >> 5) + (d.bwd ? (d.CinP_b >> 4) * (d.CoutP_b >> 5) : 0);
    }
    a.prefix[a.n] = run;
    hipLaunchKernelGGL(pack_weights_flat_bf16_kernel, dim3(dx_cdiv(run, 4)), dim3(256), 0, s, a);
  }
  DX_LAUNCH_CHECK("dx_pack_weights_host");
  return DX_OK;
}

// Y = epilogue(conv(X, Wp) + bias).  See ConvGemmArgs for the epilogue switches.
int dx_conv_gemm(const void* Xv, int ldx, const void* Wp, const float* bias, void* Yv, int ldy,
                 int B, int N, int Cin, int Cout, int taps, int bf16,
                 int relu, const float* post_scale, const float* post_shift,
                 const void* relu_auxv, int ld_aux, int accumulate,
                 const int* lens, int mask_rows, float out_scale, int skip_halo,
                 int x_bf16, int y_bf16, int aux_bf16, const int* rows_exist, void* stream) {
  const float* X = (const float*)Xv; float* Y = (float*)Yv; const float* relu_aux = (const float*)relu_auxv;
  DX_REQUIRE(X && Wp && Y, "dx_conv_gemm: null pointer");
  DX_REQUIRE(bf16 || !(x_bf16 || y_bf16 || aux_bf16), "dx_conv_gemm: bf16 storage needs bf16 operand mode");
  DX_REQUIRE(!x_bf16 || ((Cin % 8) == 0 && (ldx % 8) == 0), "dx_conv_gemm: bf16 input needs Cin, ldx multiples of 8");
  DX_REQUIRE(!y_bf16 || ((Cout % 4) == 0 && (ldy % 4) == 0 && !accumulate), "dx_conv_gemm: bf16 output needs Cout, ldy multiples of 4 and no accumulate");
  DX_REQUIRE(!aux_bf16 || ((Cout % 4) == 0 && (ld_aux % 4) == 0), "dx_conv_gemm: bf16 relu_aux needs Cout, ld_aux multiples of 4");
  DX_REQUIRE(skip_halo < 0 || lens, "dx_conv_gemm: skip_halo needs lens");
  DX_REQUIRE(B > 0 && N > 0 && Cin > 0 && Cout > 0, "dx_conv_gemm: bad dims B=%d N=%d Cin=%d Cout=%d", B, N, Cin, Cout);
  DX_REQUIRE(taps == 1 || taps == 3, "dx_conv_gemm: taps must be 1 or 3 (got %d)", taps);
  DX_REQUIRE((Cin % 4) == 0 && (ldx % 4) == 0 && ldx >= Cin, "dx_conv_gemm: Cin (%d) and ldx (%d) must be multiples of 4, ldx >= Cin", Cin, ldx);
  DX_REQUIRE(ldy >= Cout && ((Cout % 4) != 0 || (ldy % 4) == 0), "dx_conv_gemm: bad ldy %d for Cout %d", ldy, Cout);
  DX_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)Wp % 16) == 0 && ((uintptr_t)Y % 16) == 0, "dx_conv_gemm: pointers must be 16-byte aligned");
  DX_REQUIRE(!relu_aux || ld_aux >= Cout, "dx_conv_gemm: bad ld_aux");
  DX_REQUIRE((post_scale == nullptr) == (post_shift == nullptr), "dx_conv_gemm: post_scale/post_shift must come together");
  DX_REQUIRE(!mask_rows || lens, "dx_conv_gemm: mask_rows needs lens");
  int d[4];
  dx_pack_dims(Cout, Cin, bf16, d);
  ConvGemmArgs a{X, ldx, Wp, bias, Y, ldy, B, N, Cin, Cout, d[1], d[0], relu, post_scale, post_shift,
                 relu_aux, ld_aux, accumulate, lens, mask_rows, out_scale, skip_halo, x_bf16, y_bf16, aux_bf16, rows_exist};
  hipStream_t s = (hipStream_t)stream;
  dx_prof_begin(DX_PROF_CONV_GEMM, s);
  static const int use_ws = getenv("DX_CONV_WS") ? atoi(getenv("DX_CONV_WS")) : 1;
  static const int use_dk = getenv("DX_CONV_DK") ? atoi(getenv("DX_CONV_DK")) : 1;
  static const int ws_min_tiles = getenv("DX_CONV_WS_MIN_TILES") ? atoi(getenv("DX_CONV_WS_MIN_TILES")) : 64;
  // deep-K layers: weights straight from the fragment-major pack into registers, live tiles numbered first
  static const int dk_min_cin = getenv("DX_CONV_DK_MIN_CIN") ? atoi(getenv("DX_CONV_DK_MIN_CIN")) : 256;
  if (bf16 && use_dk && d[1] >= dk_min_cin && (Cin % 64) == 0 && (long)B * N * ldx < (1L << 31)) {
    if (x_bf16) { if (taps == 3) launch_conv_dk<3, true>(a, s); else launch_conv_dk<1, true>(a, s); }
    else { if (taps == 3) launch_conv_dk<3, false>(a, s); else launch_conv_dk<1, false>(a, s); }
  } else if (bf16 && use_ws && d[1] == 128 && (long)B * dx_cdiv(N, 128) >= ws_min_tiles) {      // short-K layers: weight-stationary persistent kernel
    if (x_bf16) { if (taps == 3) launch_conv_ws<3, true>(a, s); else launch_conv_ws<1, true>(a, s); }
    else { if (taps == 3) launch_conv_ws<3, false>(a, s); else launch_conv_ws<1, false>(a, s); }
  } else if (bf16) { if (taps == 3) launch_conv<dx_h16, 3>(a, s); else launch_conv<dx_h16, 1>(a, s); }
  else      { if (taps == 3) launch_conv<float, 3>(a, s);  else launch_conv<float, 1>(a, s); }
  dx_prof_end(DX_PROF_CONV_GEMM, s);
  DX_LAUNCH_CHECK("dx_conv_gemm");
  return DX_OK;
}

// G (fp32, caller-initialised, the PARAMETER's own layout (Cout, Cin, taps)) += dY^T * shift(X): a weight gradient, or a
// pre-zeroed / running `.grad` view to accumulate into.
int dx_conv_wgrad(const void* dY, int ldy, const void* X, int ldx, float* G,
                  int B, int N, int Cin, int Cout, int taps, const int* lens, int skip_halo,
                  int bf16, int dy_bf16, int x_bf16, float* dbias, void* stream) {
  DX_REQUIRE(dY && X && G, "dx_conv_wgrad: null pointer");
  DX_REQUIRE(bf16 || !(dy_bf16 || x_bf16), "dx_conv_wgrad: bf16 storage needs bf16 operand mode");
  const bool bf16_ok = (Cin % 8) == 0 && (ldx % 8) == 0 && (Cout % 8) == 0 && (ldy % 8) == 0;
  DX_REQUIRE(bf16_ok || !(dy_bf16 || x_bf16), "dx_conv_wgrad: bf16-stored operands need Cin/Cout/ld multiples of 8 (Cin=%d Cout=%d)", Cin, Cout);
  if (bf16 && bf16_ok) {   // odd tiny shapes (speaker logits) take the exact f32 kernel below
    DX_REQUIRE(B > 0 && N > 0 && Cin > 0 && Cout > 0 && (taps == 1 || taps == 3), "dx_conv_wgrad: bad dims");
    DX_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)dY % 16) == 0, "dx_conv_wgrad: pointers must be 16-byte aligned");
    DX_REQUIRE(skip_halo < 0 || lens, "dx_conv_wgrad: skip_halo needs lens");
    static const int use_wide = getenv("DX_WGRAD_WIDE") ? atoi(getenv("DX_WGRAD_WIDE")) : 1;
    // 512-thread workgroups with 128 input channels each: measured better only where the output is large enough that few token
    // slices are needed anyway (prenet 1024 x 1024: 308 -> 239 us); the 128-wide layers stay on the narrow form (59 vs 67 us)
    const bool wide = use_wide && (Cin % 128) == 0 && dx_cdiv(Cout, TILE) * dx_cdiv(Cin, 64) >= 64;
    const int tiles = dx_cdiv(Cout, TILE) * dx_cdiv(Cin, wide ? 128 : 64);
    const int total_chunks = B * dx_cdiv(N, wb_bk(taps, dy_bf16 != 0, x_bf16 != 0));
    // split-K partials are fp32 atomics (~1.3 TB/s chip-wide), so the split is sized by atomic traffic, not by "as many as fit";
    // per-kernel rocprof sweeps (r01-g build): k = 3 layers 192 / 256 / 320 / 384 blocks -> 50.0 / 47.6 / 48.8 / 50.2 us,
    // k = 1 layers 96 / 128 / 160 / 192 / 256 / 384 -> 24.9 / 20.9 / 19.5 / 18.8 / 19.5 / 22.1 us
    static const int target_blocks = getenv("DX_WGRAD_BLOCKS") ? atoi(getenv("DX_WGRAD_BLOCKS")) : 256;
    static const int target_wide = getenv("DX_WGRAD_BLOCKS_WIDE") ? atoi(getenv("DX_WGRAD_BLOCKS_WIDE")) : 256;   // one per CU
    static const int target_k1 = getenv("DX_WGRAD_BLOCKS_K1") ? atoi(getenv("DX_WGRAD_BLOCKS_K1")) : 192;
    const int ksplit = std::max(1, std::min(total_chunks, dx_cdiv(wide ? target_wide : (taps == 1 ? target_k1 : target_blocks), tiles)));
    WgradBatchArgs a{};
    a.job[0] = WgradJobDev{dY, X, G, dbias, lens, ldy, ldx, B, N, Cin, Cout, skip_halo, 0};
    a.ksplit = ksplit;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(tiles, 1, ksplit);
    dx_prof_begin(DX_PROF_WGRAD_GEMM, s);
#define DX_WG_LAUNCH(TAPS_, CIW_)                                                                                        \
    if (dy_bf16 && x_bf16) hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS_, true, true, CIW_>), grid, dim3(CIW_ * 128), 0, s, a);   \
    else if (dy_bf16) hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS_, true, false, CIW_>), grid, dim3(CIW_ * 128), 0, s, a);       \
    else if (x_bf16) hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS_, false, true, CIW_>), grid, dim3(CIW_ * 128), 0, s, a);        \
    else hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS_, false, false, CIW_>), grid, dim3(CIW_ * 128), 0, s, a);
    if (wide) { if (taps == 3) { DX_WG_LAUNCH(3, 4) } else { DX_WG_LAUNCH(1, 4) } }
    else { if (taps == 3) { DX_WG_LAUNCH(3, 2) } else { DX_WG_LAUNCH(1, 2) } }
#undef DX_WG_LAUNCH
    dx_prof_end(DX_PROF_WGRAD_GEMM, s);
    DX_LAUNCH_CHECK("dx_conv_wgrad(bf16)");
    return DX_OK;
  }
  DX_REQUIRE(skip_halo < 0 || lens, "dx_conv_wgrad: skip_halo needs lens");
  DX_REQUIRE(B > 0 && N > 0 && Cin > 0 && Cout > 0 && (taps == 1 || taps == 3), "dx_conv_wgrad: bad dims");
  DX_REQUIRE((Cin % 4) == 0 && (ldx % 4) == 0 && (Cout % 4) == 0 && (ldy % 4) == 0, "dx_conv_wgrad: Cin/Cout/ld must be multiples of 4 (Cin=%d Cout=%d)", Cin, Cout);
  DX_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)dY % 16) == 0, "dx_conv_wgrad: pointers must be 16-byte aligned");
  const int tiles = dx_cdiv(Cout, TILE) * dx_cdiv(Cin, TILE);
  const int total_chunks = B * dx_cdiv(N, WG_BK);
  int ksplit = std::max(1, std::min(total_chunks, dx_cdiv(1024, tiles * taps)));
  WgradArgs a{(const float*)dY, ldy, (const float*)X, ldx, G, B, N, Cin, Cout, ksplit, lens, skip_halo, dbias};
  hipStream_t s = (hipStream_t)stream;
  dx_prof_begin(DX_PROF_WGRAD_GEMM, s);
  if (taps == 3) hipLaunchKernelGGL(wgrad_kernel<3>, dim3(tiles, taps, ksplit), dim3(256), 0, s, a);
  else           hipLaunchKernelGGL(wgrad_kernel<1>, dim3(tiles, taps, ksplit), dim3(256), 0, s, a);
  dx_prof_end(DX_PROF_WGRAD_GEMM, s);
  DX_LAUNCH_CHECK("dx_conv_wgrad");
  return DX_OK;
}

// Several weight gradients of one kind in ONE launch (see WgradBatchArgs): bf16 operand mode, every job with Cin % 128 == 0 (the wide
// tile), same taps and the same operand storage; B, N, lens, channels, leading dimensions and the halo rule are per job.
struct DxWgradJob {       // mirrors include/daft_exprt_hip.h
  const void* dY; const void* X; float* G; float* dbias; const int* lens;
  int ldy, ldx, B, N, Cin, Cout, skip_halo, reserved;
};
int dx_conv_wgrad_batched(const void* jobs_, int njobs, int taps, int dy_bf16, int x_bf16, void* stream) {
  const DxWgradJob* jobs = reinterpret_cast<const DxWgradJob*>(jobs_);
  DX_REQUIRE(jobs && njobs > 0 && njobs <= WG_MAX_JOBS, "dx_conv_wgrad_batched: 1..%d jobs per launch (got %d)", WG_MAX_JOBS, njobs);
  DX_REQUIRE(taps == 1 || taps == 3, "dx_conv_wgrad_batched: taps must be 1 or 3");
  WgradBatchArgs a{};
  int tiles = 0, tile_jobs = 0, min_chunks = 0x7fffffff;
  for (int i = 0; i < njobs; ++i) {
    const DxWgradJob& j = jobs[i];
    DX_REQUIRE(j.dY && j.X && j.G, "dx_conv_wgrad_batched: job %d: null pointer", i);
    DX_REQUIRE(j.B > 0 && j.N > 0 && j.Cin > 0 && j.Cout > 0, "dx_conv_wgrad_batched: job %d: bad dims", i);
    DX_REQUIRE((j.Cin % 128) == 0 && (j.Cout % 8) == 0 && (j.ldx % 8) == 0 && (j.ldy % 8) == 0,
               "dx_conv_wgrad_batched: job %d: Cin must be a multiple of 128, Cout / ld* of 8 (Cin=%d Cout=%d)", i, j.Cin, j.Cout);
    DX_REQUIRE(((uintptr_t)j.X % 16) == 0 && ((uintptr_t)j.dY % 16) == 0, "dx_conv_wgrad_batched: job %d: pointers must be 16-byte aligned", i);
    DX_REQUIRE(j.skip_halo < 0 || j.lens, "dx_conv_wgrad_batched: job %d: skip_halo needs lens", i);
    a.job[i] = WgradJobDev{j.dY, j.X, j.G, j.dbias, j.lens, j.ldy, j.ldx, j.B, j.N, j.Cin, j.Cout, j.skip_halo, 0};
    const int t = dx_cdiv(j.Cout, TILE) * (j.Cin / 128);
    tiles = std::max(tiles, t);
    tile_jobs += t;
    min_chunks = std::min(min_chunks, j.B * dx_cdiv(j.N, wb_bk(taps, dy_bf16 != 0, x_bf16 != 0)));
  }
  // One 512-thread workgroup per CU and round; a launch of many layers runs several rounds (24 layers: 768 workgroups at the same 4-way
  // split), so the atomic epilogues of the early finishers run under the chunk loops of the next round instead of at the end of a launch.
  static const int per_round = getenv("DX_WGRAD_BLOCKS_BATCH") ? atoi(getenv("DX_WGRAD_BLOCKS_BATCH")) : 256;
  const int rounds = std::max(1, (tile_jobs * 4 + per_round / 2) / per_round);
  a.ksplit = std::max(1, std::min(min_chunks, rounds * per_round / tile_jobs));
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(tiles, njobs, a.ksplit);
  static const int grp_env = getenv("DX_WGRAD_XCD_GROUP") ? atoi(getenv("DX_WGRAD_XCD_GROUP")) : 1;
  a.xcd_group = grp_env && tiles == 8 && (njobs * a.ksplit) % 8 == 0;
#ifdef DX_WG_STAMPS
  a.stamps = g_wgrad_stamps;
#endif
  dx_prof_begin(DX_PROF_WGRAD_GEMM, s);
#define DX_WG_LAUNCH(TAPS_)                                                                                            \
  if (dy_bf16 && x_bf16) hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS_, true, true, 4>), grid, dim3(512), 0, s, a);     \
  else if (dy_bf16) hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS_, true, false, 4>), grid, dim3(512), 0, s, a);         \
  else if (x_bf16) hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS_, false, true, 4>), grid, dim3(512), 0, s, a);          \
  else hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS_, false, false, 4>), grid, dim3(512), 0, s, a);
  if (taps == 3) { DX_WG_LAUNCH(3) } else { DX_WG_LAUNCH(1) }
#undef DX_WG_LAUNCH
  dx_prof_end(DX_PROF_WGRAD_GEMM, s);
  DX_LAUNCH_CHECK("dx_conv_wgrad_batched");
  return DX_OK;
}

int dx_unpack_wgrad(const float* G, float* grad, int Cout, int Cin, int taps, int accumulate, void* stream) {
  DX_REQUIRE(G && grad && Cout > 0 && Cin > 0 && taps > 0, "dx_unpack_wgrad: bad arguments");
  const size_t n = (size_t)Cout * Cin * taps;
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, G, grad, Cout, Cin, taps, accumulate);
  DX_LAUNCH_CHECK("dx_unpack_wgrad");
  return DX_OK;
}

// out[c] += sum_rows X[row][c]; out is caller-initialised (bias gradients).
int dx_colsum(const void* X, int ldx, float* out, long rows, int C, int x_bf16, void* stream) {
  DX_REQUIRE(X && out && rows > 0 && C > 0 && ldx >= C, "dx_colsum: bad arguments");
  const int rpb = 256;
  dim3 grid(dx_cdiv(C, 256), (unsigned)((rows + rpb - 1) / rpb));
  if (x_bf16) hipLaunchKernelGGL(colsum_kernel<dx_h16>, grid, dim3(256), 0, (hipStream_t)stream, (const dx_h16*)X, ldx, out, rows, C, rpb);
  else hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)X, ldx, out, rows, C, rpb);
  DX_LAUNCH_CHECK("dx_colsum");
  return DX_OK;
}

}  // extern "C"

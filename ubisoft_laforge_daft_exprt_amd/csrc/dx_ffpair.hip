// Fused position-wise conv feed-forward pair, bf16 MFMA operands, gfx950.
//
//   Y[b,n,:]  = bias_b + sum_tap  Wb[tap] . Hm[b, n+tap-1, :]                 (128 <- F channels, k = 3)
//   Hm[b,m,:] = mid( bias_a + sum_tap Wa[tap] . X[b, m+tap-1, :] )   for 0 <= m < N, zero outside   (F <- 128 channels, k = 3)
//
// forward  (reference model.py:206-217, PositionWiseConvFF):  X = LayerNorm output, Wa = conv1, mid = ReLU, Wb = conv2;
//           Hm is ALSO written to HBM (bf16) because the weight gradients of both convs need it
// backward (input-gradient chain of the same pair):            X = dL/d(conv2 out), Wa = conv2 transposed + flipped,
//           mid = "zero where the forward hidden activation was <= 0" (aux = the stored Hm), Wb = conv1 transposed + flipped,
//           Y accumulates into the residual-branch gradient; the masked hidden gradient is written for the conv1 weight gradient
//
// Why fused: as two launches the F = 1024-wide hidden tensor goes HBM -> LDS again for the second conv (16.6 KB per 64-deep
// stage per workgroup), and both launches are short-K / latency-bound shells around a few thousand MFMAs.
// Here one workgroup owns 126 output tokens for its whole life (49 k cycles of MFMA per SIMD) and the hidden tensor is consumed
// from LDS in 128-channel slices while the next slice is being produced:
//
//   512 threads = 4 PRODUCER waves + 4 CONSUMER waves; SIMD s hosts producer s and consumer s, so each matrix pipe is shared by
//   one wave of each role (the pair alternates naturally: one multiplies while the other waits on LDS / converts / stores).
//   producer w, slice f:  Hm[:, f*128 + 32 w + (0..31)] for the tile's 128 hidden rows (126 + one halo row each side):
//                         192 MFMA (16x16x32), epilogue -> bf16 -> LDS slice buffer f & 1
//   consumer w, slice f:  acc[128 tokens][32 w + (0..31)] += Wb[:, slice f] . Hm slice: 192 MFMA, accumulators live across slices
//   one barrier per slice; the slice that was just completed is copied LDS -> HBM as full 128-byte row segments by all waves.
//   Weights never touch LDS: both packs are fragment-major (dx_gemm.hip, wb_off), a wave fetches an A fragment with ONE
//   contiguous 1 KB load, three 16-MFMA steps ahead (L2-resident: every workgroup streams the same 1.5 MB).
//   The 130-row activation tile (126 + 2 halo rows each side) is staged once.
// Token tile = 126, not 128: the hidden rows a tile needs are then exactly 128 = 8 MFMA column tiles (a 128-token tile would
// need 130 rows = 9 column tiles, 12 % wasted matrix work in the first conv).
#include "dx_common.h"
#include <algorithm>

namespace {

constexpr int FP_TOK = 126;      // output tokens per workgroup
constexpr int FP_HR = 130;       // rows of an LDS activation image (x: 130 used; hidden: 128 used + 2 zero rows)
constexpr int FP_IMG = 2 * FP_HR * 128;   // bytes of one image: [2 chunks of 64 channels][130 rows][128 B]
constexpr int FP_MAX_B = 1024;

struct FFPairArgs {
  const __bf16* X; int ldx;
  const __bf16* Wa; const __bf16* Wb;
  const float* bias_a; const float* bias_b;
  const __bf16* aux; int ld_aux;
  __bf16* H; int ldh;
  float* Y; int ldy;
  int B, N, F;
  int relu_mid, accumulate;
  const int* lens; int skip_halo;
};

__device__ __forceinline__ int fp_lds_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

__device__ __forceinline__ uint2 fp_pack4(float a, float b, float c, float d) {
  bf16x4 h;
  h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
  return __builtin_bit_cast(uint2, h);
}

// Keeps a value out of loop-invariant code motion: the addresses derived from it are recomputed (a handful of VALU
// instructions) where they are used instead of being hoisted out of the slice loop and held in registers across it - hoisted,
// the copy-out / epilogue / mask addresses pushed the kernel past the 256 registers two waves per SIMD allow (51 spilled).
__device__ __forceinline__ int fp_opaque(int x) { asm volatile("" : "+v"(x)); return x; }

#define FP_MMA(W, X, C) C = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, W), __builtin_bit_cast(bf16x8, X), C, 0, 0, 0);

template <bool AUX>
__global__ __launch_bounds__(512) void ff_pair_kernel(const FFPairArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Xs = smem;
  unsigned char* const Hs0 = smem + FP_IMG;
  unsigned char* const Hs1 = smem + 2 * FP_IMG;
  __shared__ int pre_s[FP_MAX_B + 1];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, wq = wave & 3;          // role 0: producer (first conv), 1: consumer (second conv)
  const int r = lane & 15, g = lane >> 4;
  const int tiles_n = (a.N + FP_TOK - 1) / FP_TOK;
  const int nslices = a.F >> 7;

  // ---- workgroup -> token tile: live tiles are numbered first (XCD round-robin then spreads them evenly), the remaining
  //      workgroups zero-fill the padding tiles ------------------------------------------------------------------------------
  int b, n0;
  bool live = true;
  if (a.skip_halo >= 0 && a.B <= FP_MAX_B) {
    if (wave == 0) {
      int run = 0;
      for (int base = 0; base < a.B; base += 64) {
        const int i = base + lane;
        const int cnt = i < a.B ? min(tiles_n, max(0, (min(a.lens[i] + a.skip_halo, a.N) + FP_TOK - 1) / FP_TOK)) : 0;
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
    int w = blockIdx.x, lo = 0, hi = a.B;
    live = w < nlive;
    if (!live) w -= nlive;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      const int key = live ? pre_s[mid] : mid * tiles_n - pre_s[mid];
      if (key <= w) lo = mid; else hi = mid;
    }
    b = lo;
    const int live_b = pre_s[b + 1] - pre_s[b];
    n0 = (live ? w - pre_s[b] : live_b + (w - (b * tiles_n - pre_s[b]))) * FP_TOK;
  } else {
    b = blockIdx.x / tiles_n;
    n0 = (blockIdx.x - b * tiles_n) * FP_TOK;
    if (a.skip_halo >= 0) live = n0 < a.lens[b] + a.skip_halo;
  }

  b = __builtin_amdgcn_readfirstlane(b);              // workgroup-uniform by construction: keep everything derived from them scalar
  n0 = __builtin_amdgcn_readfirstlane(n0);

  if (!live) {                                        // padding beyond the halo: nobody reads it with a non-zero weight; keep it defined
    const int rows = min(FP_TOK, a.N - n0);
    const int hu = a.F >> 3;                          // 16-byte units per hidden row
    for (int u = tid; u < rows * hu; u += 512) {
      const int row = u / hu, q = u - row * hu;
      *reinterpret_cast<f32x4*>(a.H + ((size_t)b * a.N + n0 + row) * a.ldh + q * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!a.accumulate)
      for (int u = tid; u < rows * 32; u += 512) {
        const int row = u >> 5, q = u & 31;
        *reinterpret_cast<f32x4*>(a.Y + ((size_t)b * a.N + n0 + row) * a.ldy + q * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    return;
  }

  // ---- weight fragment stream of this wave: step s of slice f = (tap = s / 4, ks = s % 4), two fragments (row blocks i = 0, 1) ----
  // producer: Wa pack [3][F rows][128 k]:  block ((tap * F/16 + f*8 + 2 wq + i) * 4 + ks)
  // consumer: Wb pack [3][128 rows][F k]:  block ((tap * 8 + 2 wq + i) * F/32 + f*4 + ks)
  const __bf16* const wbase = role == 0 ? a.Wa + (size_t)(2 * wq) * 4 * 512
                                        : a.Wb + (size_t)(2 * wq) * (a.F >> 5) * 512;       // wave-uniform: scalar registers
  const unsigned lane16 = lane * 16;                  // + a 32-bit per-lane byte offset: global_load with a scalar base
  const int w_tap = role == 0 ? (a.F >> 4) * 4 * 512 : 8 * (a.F >> 5) * 512;      // elements between taps
  const int w_i = role == 0 ? 4 * 512 : (a.F >> 5) * 512;                          // between the two row blocks
  const int w_slice = role == 0 ? 8 * 4 * 512 : 4 * 512;                           // between slices
  const int total_steps = nslices * 12;
  auto w_ptr = [&](int gs) {                          // global step index -> fragment (i = 0) address; past the end: re-read the last
    gs = min(gs, total_steps - 1);
    const int f = gs / 12, s = gs - f * 12;
    return wbase + (size_t)f * w_slice + (s >> 2) * w_tap + (s & 3) * 512;
  };
  f32x4 wr[4][2];                                     // ring of 4 steps
#define FP_WLOAD(SLOT, GS)                                                                                           \
  {                                                                                                                  \
    const char* p_ = reinterpret_cast<const char*>(w_ptr(GS));                                                       \
    wr[SLOT][0] = *reinterpret_cast<const f32x4*>(p_ + lane16);                                                      \
    wr[SLOT][1] = *reinterpret_cast<const f32x4*>(p_ + (size_t)w_i * 2 + lane16);                                    \
  }
  FP_WLOAD(0, 0) FP_WLOAD(1, 1) FP_WLOAD(2, 2) FP_WLOAD(3, 3)

  // ---- stage the activation tile: rows p = 0..129 <-> n = n0 - 2 + p, zero outside [0, N) ---------------------------------
  {
    f32x4 xr[5];
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int u = tid + it * 512;
      const int row = u >> 4, q = u & 15;
      const int n = n0 - 2 + row;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u < FP_HR * 16 && n >= 0 && n < a.N) v = *reinterpret_cast<const f32x4*>(a.X + ((size_t)b * a.N + n) * a.ldx + q * 8);
      xr[it] = v;
    }
    // rows 128, 129 of both hidden images stay zero for the whole kernel (the last two MFMA columns of the second conv read them)
    if (tid < 64) {
      const int img = tid >> 5, rr = 128 + ((tid >> 4) & 1), q = tid & 15;
      *reinterpret_cast<f32x4*>((img ? Hs1 : Hs0) + (q >> 3) * (FP_HR * 128) + fp_lds_off(rr, q & 7)) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int u = tid + it * 512;
      const int row = u >> 4, q = u & 15;
      if (u < FP_HR * 16) *reinterpret_cast<f32x4*>(Xs + (q >> 3) * (FP_HR * 128) + fp_lds_off(row, q & 7)) = xr[it];
    }
  }
  __syncthreads();

  // fragment read offsets inside an image: row = 16 j + r + tap, slot = (ks & 1) * 4 + g; (row & 7) does not depend on j, so the
  // swizzled offset is one value per (tap, ks & 1) plus compile-time constants for j and the channel chunk
  int foff[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) foff[t][hh] = fp_lds_off(r + t, hh * 4 + g);
  const int len_cols = min(FP_TOK, a.N - n0);         // output columns of this tile that exist

  // one slice of one role: 12 steps of 16 MFMA; B fragments from `img`, A fragments from the register ring (refilled three steps
  // ahead, across slice boundaries)
#define FP_RD8(DST, IMG, S)                                                                                          \
  {                                                                                                                  \
    const unsigned char* const bp_ = (IMG) + (((S) & 3) >> 1) * (FP_HR * 128) + foff[(S) >> 2][(S) & 1];             \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) DST[j] = *reinterpret_cast<const float4*>(bp_ + j * 2048);         \
  }
#define FP_STEP(CUR, NXT, IMG, GS0, S, AUXPF, FRESH)                                                                 \
  {                                                                                                                  \
    if ((S) + 1 < 12) FP_RD8(NXT, IMG, (S) + 1)                 /* next step's B fragments, under this step's MFMAs */ \
    const f32x4 w0 = wr[(S) & 3][0], w1 = wr[(S) & 3][1];                                                            \
    if ((FRESH) && (S) == 0) {                                  /* a producer slice starts from a literal zero C operand */ \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { acc[0][j] = f32x4{0.f, 0.f, 0.f, 0.f}; FP_MMA(w0, CUR[j], acc[0][j]) } \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f}; FP_MMA(w1, CUR[j], acc[1][j]) } \
    } else {                                                                                                         \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { FP_MMA(w0, CUR[j], acc[0][j]) }                                \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) { FP_MMA(w1, CUR[j], acc[1][j]) }                                \
    }                                                                                                                \
    FP_WLOAD((S) & 3, (GS0) + (S) + 4)                          /* the ring slot just consumed: four steps ahead */  \
    if ((S) == 8) { AUXPF }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
#define FP_SLICE_STEPS(IMG, GS0, AUXPF, FR)                                                                          \
  {                                                                                                                  \
    float4 xa[8], xb[8];                                                                                             \
    FP_RD8(xa, IMG, 0)                                                                                               \
    FP_STEP(xa, xb, IMG, GS0, 0, AUXPF, FR) FP_STEP(xb, xa, IMG, GS0, 1, AUXPF, FR) FP_STEP(xa, xb, IMG, GS0, 2, AUXPF, FR)   \
    FP_STEP(xb, xa, IMG, GS0, 3, AUXPF, FR) FP_STEP(xa, xb, IMG, GS0, 4, AUXPF, FR) FP_STEP(xb, xa, IMG, GS0, 5, AUXPF, FR)   \
    FP_STEP(xa, xb, IMG, GS0, 6, AUXPF, FR) FP_STEP(xb, xa, IMG, GS0, 7, AUXPF, FR) FP_STEP(xa, xb, IMG, GS0, 8, AUXPF, FR)   \
    FP_STEP(xb, xa, IMG, GS0, 9, AUXPF, FR) FP_STEP(xa, xb, IMG, GS0, 10, AUXPF, FR) FP_STEP(xb, xa, IMG, GS0, 11, AUXPF, FR) \
  }

  // -- copy-out of the slice completed in the previous iteration (hidden rows 1..126 of the image <-> tokens n0 .. n0+125):
  //    full 128-byte row segments, all eight waves
#define FP_COPY_OUT(IT)                                                                                              \
  if ((IT) >= 1) {                                                                                                   \
    const unsigned char* img_ = (((IT) - 1) & 1) ? Hs1 : Hs0;                                                        \
    const int f0c_ = ((IT) - 1) << 7;                                                                                \
    const int tid_ = fp_opaque(tid);                                                                                 \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                  \
      const int u = tid_ + k * 512;                                                                                  \
      const int row = u >> 4, q = u & 15;                                                                            \
      if (row < len_cols) {                                                                                          \
        const f32x4 v = *reinterpret_cast<const f32x4*>(img_ + (q >> 3) * (FP_HR * 128) + fp_lds_off(row + 1, q & 7)); \
        *reinterpret_cast<f32x4*>(a.H + ((size_t)b * a.N + n0 + row) * a.ldh + f0c_ + q * 8) = v;                    \
      }                                                                                                              \
    }                                                                                                                \
  }

  // The two roles run SEPARATE loops (same number of barriers in each): one loop with a role branch inside made the register
  // allocator carry the accumulators of both paths through common phis and rotate them through 64 extra registers.
  if (role == 0) {
    for (int it = 0; it <= nslices; ++it) {
      FP_COPY_OUT(it)
      if (it < nslices) {
        const int f = it, f0 = f << 7;
        // backward: the sign mask of the forward activation, 4 channels x 16 (i, j) positions per lane; requested during the
        // matrix steps (clamped rows, unconditional) so that the epilogue does not start with sixteen dependent round trips
        bf16x4 av[2][8];
        f32x4 acc[2][8];
        FP_SLICE_STEPS(Xs, f * 12,
          if constexpr (AUX) {
            const int r2_ = fp_opaque(r);
            const int g2_ = fp_opaque(g);
            _Pragma("unroll") for (int i = 0; i < 2; ++i)
              _Pragma("unroll") for (int j = 0; j < 8; ++j) {
                const int n = min(max(n0 - 1 + 16 * j + r2_, 0), a.N - 1);
                av[i][j] = *reinterpret_cast<const bf16x4*>(a.aux + ((size_t)b * a.N + n) * a.ld_aux + f0 + 32 * wq + 16 * i + 4 * g2_);
              }
          }, true)
        // producer epilogue: bias + ReLU (forward) or the sign mask (backward), bf16, into slice image f & 1.
        // Hidden rows outside [0, N) of the batch row are the second conv's zero padding.
        unsigned char* const out = (f & 1) ? Hs1 : Hs0;
        const int r_ = fp_opaque(r), g_ = fp_opaque(g);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = 32 * wq + 16 * i + 4 * g_;    // channel inside the slice
          float bv[4] = {0.f, 0.f, 0.f, 0.f};
          if (a.bias_a) { const float4 t = *reinterpret_cast<const float4*>(a.bias_a + f0 + c); bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[3] = t.w; }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int q = 16 * j + r_;
            const int n = n0 - 1 + q;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = acc[i][j][e] + bv[e];
              if (a.relu_mid) v[e] = fmaxf(v[e], 0.f);
              if constexpr (AUX) { if (!((float)av[i][j][e] > 0.f)) v[e] = 0.f; }
            }
            if (n < 0 || n >= a.N) { v[0] = v[1] = v[2] = v[3] = 0.f; }
            *reinterpret_cast<uint2*>(out + (c >> 6) * (FP_HR * 128) + fp_lds_off(q, (c & 63) >> 3) + ((g_ & 1) << 3)) = fp_pack4(v[0], v[1], v[2], v[3]);
          }
        }
      }
      __syncthreads();
    }
  } else {
    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it <= nslices; ++it) {
      FP_COPY_OUT(it)
      if (it >= 1) {
        const int f = it - 1;
        const unsigned char* const img = (f & 1) ? Hs1 : Hs0;
        FP_SLICE_STEPS(img, f * 12, , false)
      }
      __syncthreads();
    }
    // ---- consumer epilogue: lane holds output channels co .. co+3 of token n0 + 16 j + r ---------------------------------
    // accumulate: all sixteen old values are requested first (clamped addresses, unconditional) - a load / add / store per
    // position under its own bounds check made sixteen dependent round trips of it
    f32x4 old[2][8];
    if (a.accumulate) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int col = min(16 * j + r, len_cols - 1);
          old[i][j] = *reinterpret_cast<const f32x4*>(a.Y + ((size_t)b * a.N + n0 + col) * a.ldy + 32 * wq + 16 * i + 4 * g);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int co = 32 * wq + 16 * i + 4 * g;
      f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.bias_b) bv = *reinterpret_cast<const f32x4*>(a.bias_b + co);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = 16 * j + r;
        f32x4 o = acc[i][j] + bv;
        if (a.accumulate) o += old[i][j];
        if (col < len_cols) *reinterpret_cast<f32x4*>(a.Y + ((size_t)b * a.N + n0 + col) * a.ldy + co) = o;
      }
    }
  }
#undef FP_COPY_OUT
#undef FP_SLICE_STEPS
#undef FP_STEP
#undef FP_RD8

#undef FP_WLOAD
}

#undef FP_MMA

}  // namespace

extern "C" {

// One launch for conv(k=3, 128 -> F) -> mid -> conv(k=3, F -> 128) on channels-last bf16 activations (see the header of this file).
// Wa / Wb: fragment-major bf16 packs written by dx_pack_weights (forward pair: conv1.fwd / conv2.fwd; input-gradient pair:
// conv2.bwd / conv1.bwd).  H [B][N][F] bf16 receives the mid activation; Y [B][N][128] fp32 the result (+= if accumulate).
// aux (optional, bf16 [B][N][F]): mid = "zero where aux <= 0" instead of / after ReLU.  skip_halo: token tiles that start at or
// beyond min(lens[b] + skip_halo, N) are padding nobody reads: zero-filled, not computed.
int dx_ff_pair(const void* X, int ldx, const void* Wa, const void* Wb, const float* bias_a, const float* bias_b,
               const void* aux, int ld_aux, void* H, int ldh, float* Y, int ldy,
               int B, int N, int F, int relu_mid, int accumulate, const int* lens, int skip_halo, void* stream) {
  DX_REQUIRE(X && Wa && Wb && H && Y, "dx_ff_pair: null pointer");
  DX_REQUIRE(B > 0 && N > 0 && F >= 128 && (F % 128) == 0, "dx_ff_pair: bad dims B=%d N=%d F=%d (F must be a multiple of 128)", B, N, F);
  DX_REQUIRE(ldx >= 128 && (ldx % 8) == 0 && ldh >= F && (ldh % 8) == 0 && ldy >= 128 && (ldy % 4) == 0, "dx_ff_pair: bad leading dimensions");
  DX_REQUIRE(!aux || (ld_aux >= F && (ld_aux % 4) == 0), "dx_ff_pair: bad ld_aux");
  DX_REQUIRE(skip_halo < 0 || lens, "dx_ff_pair: skip_halo needs lens");
  DX_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)Wa % 16) == 0 && ((uintptr_t)Wb % 16) == 0 && ((uintptr_t)H % 16) == 0 &&
             ((uintptr_t)Y % 16) == 0 && ((uintptr_t)aux % 8) == 0, "dx_ff_pair: pointers must be 16-byte aligned");
  FFPairArgs a{(const __bf16*)X, ldx, (const __bf16*)Wa, (const __bf16*)Wb, bias_a, bias_b, (const __bf16*)aux, ld_aux,
               (__bf16*)H, ldh, Y, ldy, B, N, F, relu_mid, accumulate, lens, skip_halo};
  const size_t smem = 3 * (size_t)FP_IMG;
  static bool configured = false;
  if (!configured) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_pair_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_pair_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  hipStream_t s = (hipStream_t)stream;
  dx_prof_begin(DX_PROF_CONV_GEMM, s);
  if (aux) hipLaunchKernelGGL(ff_pair_kernel<true>, dim3(B * dx_cdiv(N, FP_TOK)), dim3(512), smem, s, a);
  else hipLaunchKernelGGL(ff_pair_kernel<false>, dim3(B * dx_cdiv(N, FP_TOK)), dim3(512), smem, s, a);
  dx_prof_end(DX_PROF_CONV_GEMM, s);
  DX_LAUNCH_CHECK("dx_ff_pair");
  return DX_OK;
}

}  // extern "C"

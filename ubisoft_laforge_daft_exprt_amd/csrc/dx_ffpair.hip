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
//                         192 MFMA (16x16x32), epilogue -> bf16 -> LDS slice image f & 1
//   consumer w, slice f:  acc[128 tokens][32 w + (0..31)] += Wb[:, slice f] . Hm slice: 192 MFMA, accumulators live across slices
//   one barrier per slice; the slice completed in the previous iteration is copied LDS -> HBM as full 128-byte row segments, one
//   piece per pair of matrix steps (so it hides under them).
//   Weights never touch LDS: both packs are fragment-major (dx_gemm.hip, wb_off), a wave fetches an A fragment with ONE
//   contiguous 1 KB load, four 16-MFMA steps ahead (L2-resident: every workgroup streams the same 1.5 MB).
//   The 130-row activation tile (126 + 2 halo rows each side) is staged once.
// Token tile = 126, not 128: the hidden rows a tile needs are then exactly 128 = 8 MFMA column tiles (a 128-token tile would
// need 130 rows = 9 column tiles, 12 % wasted matrix work in the first conv).
//
// Row passes folded into the two ends of the kernel (a 126 x 128 tile owns whole rows of the 128-wide streams):
//   forward epilogue (dx_ff_pair_ln)      dropout + residual + the block's SECOND LayerNorm + FiLM + mask on the output tile (model.py:225-233)
//   backward prologue (dx_ff_block_bwd)   the backward of that LayerNorm COMPUTES the activation tile (rows + halo) instead of loading it
//   backward epilogue (dx_ff_pair_lnbwd,  the backward of the block's FIRST LayerNorm on the output tile, then - dx_ff_block_bwd - the
//                      dx_ff_block_bwd)   attention out-projection's input-gradient GEMM on the resulting 16-bit rows
// Each removes a launch and an HBM round trip of a 512 B/token tensor; they use registers only after / before the slice loop.
#include "dx_common.h"
#include <algorithm>

namespace {

// NJ = MFMA column tiles (16 tokens each) of a workgroup's token tile.  NJ = 8 (126 output tokens, 100 KB of LDS, 256 registers: ONE workgroup
// per CU) was the round-2 kernel: 29 % of such a workgroup's life is fill (the first slice is produced with the consumers idle), drain and the
// row-pass epilogues, during which the CU's matrix pipes idle.  NJ = 4 (62 output tokens, 51 KB, <= 128 registers) lets TWO workgroups share
// a CU: four waves per SIMD, and one workgroup's fill / drain / epilogue runs beside the other's slice loop.  The price: every weight
// fragment is reused by 4 instead of 8 MFMAs, i.e. twice the L2 -> register weight traffic per FLOP (768 MB per launch), and at the
// 128-register cap the kernel spills 41-45 registers.  MEASURED (round 3, C2 frame axis, same box): NJ = 4 with two workgroups per CU is
// slower -- forward 80-93 us against 56, backward 96-101 against 68 -- so the frame axis stays on NJ = 8.  NJ = 4 (built at 256 registers,
// no spills, one workgroup per CU) serves the SHORT batches instead: the symbol axis (48 x <= 120 symbols) is 48 tiles of 126 tokens but
// 96 of 62, and the fused forms then replace four (forward) and five (backward) latency-bound launches per symbol-level block.
template <int NJ> struct FP {
  static constexpr int NROW = NJ * 16;       // hidden rows of a tile = MFMA columns of the first conv
  static constexpr int TOK = NROW - 2;       // output tokens per workgroup
  static constexpr int HR = NROW + 2;        // rows of an LDS activation image (x: HR used; hidden: NROW used + 2 zero rows)
  static constexpr int IMG = 2 * HR * 128;   // bytes of one image: [2 chunks of 64 channels][HR rows][128 B]
  static constexpr int XP = (HR * 16 + 511) / 512;     // passes of 512 threads over the HR x 16 staging units
  static constexpr int RP = (HR + 31) / 32;            // passes of 32 rows (16 lanes x 8 channels per row) over the HR rows
};
constexpr int FP_TOK = FP<8>::TOK;           // (host-side helpers and the Python side's "126-token tile" refer to the NJ = 8 geometry)

struct FFPairArgs {
  const dx_h16* X; int ldx;
  const dx_h16* Wa; const dx_h16* Wb;
  const float* bias_a; const float* bias_b;
  const dx_h16* aux; int ld_aux;
  dx_h16* H; int ldh;
  float* Y; int ldy;
  int B, N, F;
  int accumulate;
  const int* lens; int skip_halo;
  const int* rows_exist;             // optional [B]: input / hidden rows n >= rows_exist[b] do not exist (see dx_gemm.hip ConvGemmArgs)
  // optional LayerNorm epilogue of the forward (ln_w != null): Y receives z = ln_res + dropout(conv output), ln_y = mask(FiLM(LN(z)))
  // -- the block's second LayerNorm (model.py:225-233) done on the output tile while it is in LDS (same arithmetic as dx_ln_fwd)
  const float* ln_res; const float* ln_w; const float* ln_b; const float* film; int ld_film;
  float* ln_y; float* ln_mean; float* ln_rstd;
  unsigned long long seed_pre; unsigned thresh_pre; float inv_keep_pre; const unsigned long long* seed_offset;
  // ... optionally followed by the NEXT block's attention in-projection on the same tile (q_w != null): q_out = ln_y x W_in^T + q_bias,
  // 16-bit [B][N][384] (q_w: the forward pack of the (384, 128) in-projection weight) -- what dx_conv_gemm(taps = 1, halo 0) writes from ln_y
  const dx_h16* q_w; const float* q_bias; dx_h16* q_out;
  // optional LayerNorm-BACKWARD epilogue of the input-gradient pair (lnb_w != null): the pair's result + the residual gradient already in Y
  // is d(loss)/d(y1) of the block's FIRST LayerNorm; the epilogue turns the tile into dz1 (written to Y), its dropped-out 16-bit copy for the
  // GEMMs (lnb_dg) and the affine gradients (lnb_dw / lnb_db, atomics) -- dx_ln_bwd with C = 128, no FiLM, halo 0, done while the tile is in LDS
  const float* lnb_z; const float* lnb_mean; const float* lnb_rstd; const float* lnb_w; const float* lnb_b;
  dx_h16* lnb_dg; float* lnb_dw; float* lnb_db;
  // ... optionally followed by the input-gradient GEMM of the attention out-projection on the same tile: DATT = lnb_dg x W_out
  // (lnb_wt: the backward pack of the (128, 128) out-projection weight; lnb_datt: 16-bit [B][N][128], what the attention backward reads as dctx)
  const dx_h16* lnb_wt; dx_h16* lnb_datt;
  // optional LayerNorm-BACKWARD prologue of the input-gradient pair (lnp_w != null; needs the epilogue above): X is not read; the tile's
  // rows of d(loss)/d(y2) (lnp_dy) go through the backward of the block's SECOND LayerNorm on their way into LDS: dz2 -> Y (the residual
  // gradient the epilogue then adds to), dropout(dz2) as 16 bits -> the LDS tile and lnp_dg (for the weight gradient of the second conv),
  // affine / FiLM gradients -> lnp_dw, lnp_db, lnp_dfilm (atomics).  dx_ln_bwd with C = 128, FiLM optional, halo 0.
  const float* lnp_dy; const float* lnp_z; const float* lnp_mean; const float* lnp_rstd; const float* lnp_w; const float* lnp_b;
  const float* lnp_film; int lnp_ld_film;
  dx_h16* lnp_dg; float* lnp_dw; float* lnp_db; float* lnp_dfilm; int lnp_ld_dfilm;
  unsigned long long lnp_seed; unsigned lnp_thresh; float lnp_inv_keep;
  // optional ReLU-sign words (forward: written; backward: read INSTEAD of aux): one dword per lane, slice and 16-channel block, bit k of it =
  // "hidden value (column tile j = NJ-1 - k/4, channel 3 - k%4 of the lane's four) was > 0"; [B * tiles][F / 128][4 waves][2][64 lanes]
  unsigned* hmask;
  int slice_skew;                    // 1: workgroup w starts its walk over the hidden slices at slice w % nslices (see the kernel)
  int dz_lds;                        // 1 (block backward with the LayerNorm prologue): dz2 waits for the epilogue in LDS behind the images instead of in Y (see the prologue)
  unsigned long long* stamps;        // diagnostic builds (-DDX_FFPAIR_STAMPS, tools/ffpair_stamps.py) only: [workgroup][role][16] s_memtime values
};

#ifdef DX_FFPAIR_STAMPS
#define FP_STAMP(K) { if (lane == 0 && wq == 0 && a.stamps) a.stamps[((size_t)blockIdx.x * 2 + role) * 16 + (K)] = __builtin_amdgcn_s_memtime(); }
#else
#define FP_STAMP(K) {}
#endif

__device__ __forceinline__ int fp_lds_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

__device__ __forceinline__ uint2 fp_pack4(float a, float b, float c, float d) {
  bf16x4 h;
  h[0] = (dx_h16)a; h[1] = (dx_h16)b; h[2] = (dx_h16)c; h[3] = (dx_h16)d;
  return __builtin_bit_cast(uint2, h);
}

// Keeps a value out of loop-invariant code motion: the addresses derived from it are recomputed (a handful of VALU
// instructions) where they are used instead of being hoisted out of the slice loop and held in registers across it - hoisted,
// the copy-out / epilogue / mask addresses pushed the kernel past the 256 registers two waves per SIMD allow (51 spilled).
__device__ __forceinline__ int fp_opaque(int x) { asm volatile("" : "+v"(x)); return x; }

__device__ __forceinline__ int fp_wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Timing ablations for tools/ffpair_stamps.py (diagnostic builds only; results are then numerically wrong on purpose):
//   DX_FP_ABL = 1: weight fragments are fetched once and reused;  2: B fragments are read from LDS once per slice and reused;
//   3: the producer epilogue is skipped;  4: 1 + 2 + no copy-out: bare MFMA steps.  (asm volatile keeps the reused values alive so nothing upstream is dead-code eliminated.)
#ifndef DX_FP_ABL
#define DX_FP_ABL 0
#endif

typedef short s16x2 __attribute__((ext_vector_type(2)));

#define FP_MMA(W, X, C) C = DX_MFMA_H16(__builtin_bit_cast(bf16x8, W), __builtin_bit_cast(bf16x8, X), C);

// AUX: backward (mid = sign mask of the stored forward activation); RELU: forward (mid = ReLU)
// MASK (backward only): the ReLU gradient mask comes from the forward's sign words (a.hmask) instead of the stored activation (a.aux) --
// a compile-time choice: with both paths in one kernel the aux prefetch registers stay allocated and the kernel spills
template <bool AUX, bool RELU, int NJ, bool MASK = false>
__global__ __launch_bounds__(512, 2) void ff_pair_kernel(const FFPairArgs a) {
  constexpr int FP_TOK = FP<NJ>::TOK, FP_HR = FP<NJ>::HR, FP_IMG = FP<NJ>::IMG, NROW = FP<NJ>::NROW, XP = FP<NJ>::XP, RP = FP<NJ>::RP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Xs = smem;
  unsigned char* const Hs0 = smem + FP_IMG;
  unsigned char* const Hs1 = smem + 2 * FP_IMG;
  // The block backward's prologue produces dz2 (the residual-branch gradient) and its epilogue adds the pair's result to it: through Y that is
  // 512 B/token written and 512 B/token read back in the two HBM-bound, matrix-idle ends of the kernel.  The rows that fit behind the three
  // images (126-token tiles: 112 of 126 rows = 56 KB, 157 KB of LDS in all; 62-token tiles: all rows) wait there instead; the rest go through Y.
  constexpr int DZ_ROWS = NJ == 8 ? 112 : FP_TOK;
  float* const dzs = (a.lnp_w && a.dz_lds) ? reinterpret_cast<float*>(smem + 3 * FP_IMG) : nullptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave >> 2, wq = wave & 3;          // role 0: producer (first conv), 1: consumer (second conv)
  const int r = lane & 15, g = lane >> 4;
  const int tiles_n = (a.N + FP_TOK - 1) / FP_TOK;
  const int nslices = a.F >> 7;
  FP_STAMP(0)

  // ---- weight fragment stream of this wave: step s of slice f = (tap = s / 4, ks = s % 4), two fragments (row blocks i = 0, 1) ----
  // producer: Wa pack [3][F rows][128 k]:  block ((tap * F/16 + f*8 + 2 wq + i) * 4 + ks)
  // consumer: Wb pack [3][128 rows][F k]:  block ((tap * 8 + 2 wq + i) * F/32 + f*4 + ks)
  const dx_h16* const wbase = role == 0 ? a.Wa + (size_t)(2 * wq) * 4 * 512
                                        : a.Wb + (size_t)(2 * wq) * (a.F >> 5) * 512;       // wave-uniform: scalar registers
  const unsigned lane16 = lane * 16;                  // + a 32-bit per-lane byte offset: global_load with a scalar base
  const int w_tap = role == 0 ? (a.F >> 4) * 4 * 512 : 8 * (a.F >> 5) * 512;      // elements between taps
  const int w_i = role == 0 ? 4 * 512 : (a.F >> 5) * 512;                          // between the two row blocks
  const int w_slice = role == 0 ? 8 * 4 * 512 : 4 * 512;                           // between slices
  const int total_steps = nslices * 12;
  // Slice ORDER: iteration `it` of a workgroup handles hidden slice (it + skew) % nslices, skew = its block index.  Every workgroup of a launch
  // streams the same 1.5 MB of weights; started together and walking the slices in the same order they all ask the XCD's L2 for the
  // SAME kilobytes at the same time, i.e. for the one or two L2 channels those addresses live in: the fragment stream ran at ~7 TB/s chip-wide
  // whatever the tile size (126-token tiles: 384 MB per launch in 56 us; 62-token tiles: 768 MB in 93 us).  Skewed, the 32 workgroups of an XCD
  // are spread over all nslices x 12 steps of the stream.  (A sum over slices in another order: fp32 rounding only; the order is a function of the
  // block index, so results stay reproducible run to run.)
  const int skew = a.slice_skew ? (int)((blockIdx.x >> 3) % (unsigned)nslices) : 0;   // (blockIdx & 7 = the XCD: neighbours ON an XCD must differ)
  auto slice_of = [&](int it) { const int f = it + skew; return f >= nslices ? f - nslices : f; };
  auto w_ptr = [&](int gs) {                          // global step index (iteration order) -> fragment (i = 0) address; past the end: re-read the last
    gs = min(gs, total_steps - 1);
    const int it_ = gs / 12, s = gs - it_ * 12;
    const int f = slice_of(it_);
    return wbase + (size_t)f * w_slice + (s >> 2) * w_tap + (s & 3) * 512;
  };
  // ring of RING steps: a step is 2 * NJ MFMAs per wave (NJ = 8: 256 cycles, four steps ahead = ~1 k cycles of cover for an L2 round trip; NJ = 4:
  // 128 cycles per step, so six steps ahead -- 12 steps per slice keep the slot of a step a compile-time constant for 4 and 6)
  constexpr int RING = NJ == 8 ? 4 : 6;
  f32x4 wr[RING][2];
#define FP_WLOAD(SLOT, GS)                                                                                           \
  {                                                                                                                  \
    const char* p_ = reinterpret_cast<const char*>(w_ptr(GS));                                                       \
    wr[SLOT][0] = *reinterpret_cast<const f32x4*>(p_ + lane16);                                                      \
    wr[SLOT][1] = *reinterpret_cast<const f32x4*>(p_ + (size_t)w_i * 2 + lane16);                                    \
  }

  // ---- workgroup -> token tile: live tiles are numbered first (XCD round-robin then spreads them evenly), the remaining
  //      workgroups zero-fill the padding tiles.  Every wave finds the tile for itself with wave scans / ballots (no LDS, no
  //      barrier: the shared-prefix + binary-search form of the other conv kernels costs ~3.2 k cycles here, stamped) ----------
  int b, n0;
  bool live = true;
  if (a.skip_halo >= 0) {
    // (one load of the lengths and one scan serve both questions -- how many live tiles are there, and which tile is mine: the dead tiles
    // before row i are (i + 1) * tiles_n minus the live ones; a second pass over lens[] was a second dependent global round trip per workgroup)
    int nlive = 0;
    int cnt0 = 0, inc0 = 0;                             // tile count of row `lane` and its inclusive scan (first 64 rows)
    for (int base = 0; base < a.B; base += 64) {
      const int i = base + lane;
      const int cnt = i < a.B ? min(tiles_n, max(0, (min(a.lens[i] + a.skip_halo, a.N) + FP_TOK - 1) / FP_TOK)) : 0;
      int inc = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (lane >= off) inc += v; }
      if (base == 0) { cnt0 = cnt; inc0 = inc; }
      nlive += __shfl(inc, 63, 64);
    }
    int target = blockIdx.x;
    live = target < nlive;
    if (!live) target -= nlive;
    int run = 0;
    b = 0; n0 = 0;
    for (int base = 0; base < a.B; base += 64) {
      const int i = base + lane;
      int cnt, inc_live;
      if (base == 0) { cnt = cnt0; inc_live = inc0; }
      else {
        cnt = i < a.B ? min(tiles_n, max(0, (min(a.lens[i] + a.skip_halo, a.N) + FP_TOK - 1) / FP_TOK)) : 0;
        inc_live = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc_live, off, 64); if (lane >= off) inc_live += v; }
      }
      const int rows_here = min(64, a.B - base);
      const int c = live ? cnt : (i < a.B ? tiles_n - cnt : 0);       // tiles of the wanted kind in this row
      const int inc = live ? inc_live : min(lane + 1, rows_here) * tiles_n - inc_live;
      const unsigned long long hit = __ballot(target < run + inc);
      if (hit) {
        const int first = __builtin_ctzll(hit);
        const int excl = __shfl(run + inc - c, first, 64), cnt_b = __shfl(cnt, first, 64);
        b = base + first;
        n0 = (live ? target - excl : cnt_b + (target - excl)) * FP_TOK;
        break;
      }
      run += __shfl(inc, 63, 64);
    }
  } else {
    b = blockIdx.x / tiles_n;
    n0 = (blockIdx.x - b * tiles_n) * FP_TOK;
  }
  FP_STAMP(1)
  b = __builtin_amdgcn_readfirstlane(b);              // workgroup-uniform by construction: keep everything derived from them scalar
  n0 = __builtin_amdgcn_readfirstlane(n0);

  if (!live) {                                        // padding beyond the halo: nobody reads it with a non-zero weight; keep it defined
    const int rows = min(FP_TOK, a.N - n0);
    const int hu = a.H ? a.F >> 3 : 0;                // 16-byte units per hidden row (no hidden output in a forward-only call)
    for (int u = tid; u < rows * hu; u += 512) {
      const int row = u / hu, q = u - row * hu;
      *reinterpret_cast<f32x4*>(a.H + ((size_t)b * a.N + n0 + row) * a.ldh + q * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a.lnb_w)
      for (int u = tid; u < rows * 16; u += 512) {        // 16-bit gradient copies: 16 x 16 bytes per row
        const int row = u >> 4, q = u & 15;
        *reinterpret_cast<f32x4*>(a.lnb_dg + ((size_t)b * a.N + n0 + row) * 128 + q * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.lnp_w) *reinterpret_cast<f32x4*>(a.lnp_dg + ((size_t)b * a.N + n0 + row) * 128 + q * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.lnb_wt) *reinterpret_cast<f32x4*>(a.lnb_datt + ((size_t)b * a.N + n0 + row) * 128 + q * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    if (a.lnp_w)                                          // Y was nobody's output before this launch: the residual gradient of padding is zero
      for (int u = tid; u < rows * 32; u += 512) {
        const int row = u >> 5, q = u & 31;
        *reinterpret_cast<f32x4*>(a.Y + ((size_t)b * a.N + n0 + row) * a.ldy + q * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    if (!a.accumulate)
      for (int u = tid; u < rows * 32; u += 512) {
        const int row = u >> 5, q = u & 31;
        *reinterpret_cast<f32x4*>(a.Y + ((size_t)b * a.N + n0 + row) * a.ldy + q * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.ln_w) *reinterpret_cast<f32x4*>(a.ln_y + ((size_t)b * a.N + n0 + row) * 128 + q * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    if (a.q_w)
      for (int u = tid; u < rows * 48; u += 512) {        // 384 16-bit values per row
        const int row = u / 48, q = u - row * 48;
        *reinterpret_cast<f32x4*>(a.q_out + ((size_t)b * a.N + n0 + row) * 384 + q * 8) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    return;
  }

  const int NL = a.rows_exist ? a.rows_exist[b] : a.N;   // rows of this batch row that exist for the two convolutions
  if (a.lnp_w) {
    // ---- LayerNorm-backward prologue: the activation tile is COMPUTED (16 lanes x 8 channels per row, 32 rows per pass) ------------
    const int q = tid & 15;
    const int len_b = a.lens ? a.lens[b] : a.N;
    float wv[8], bv[8], fg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      wv[e] = a.lnp_w[q * 8 + e]; bv[e] = a.lnp_b[q * 8 + e];
      fg[e] = a.lnp_film ? a.lnp_film[(size_t)b * a.lnp_ld_film + q * 8 + e] : 1.f;
    }
    const unsigned long long seed = a.lnp_seed + (a.seed_offset ? *a.seed_offset : 0ull);
    float gw[8], gb[8], gfg[8], gfb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) gw[e] = gb[e] = gfg[e] = gfb[e] = 0.f;
    f32x4 dyv[RP][2], zv[RP][2];
    float muv[RP], rsv[RP];
#pragma unroll
    for (int it = 0; it < RP; ++it) {                   // all global reads first
      const int row = (tid >> 4) + it * 32, n = n0 - 2 + row;
      const bool valid = row < FP_HR && n >= 0 && n < len_b && n < NL;
      const size_t grow = (size_t)b * a.N + n;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        dyv[it][hh] = valid ? *reinterpret_cast<const f32x4*>(a.lnp_dy + grow * 128 + q * 8 + hh * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        zv[it][hh] = valid ? *reinterpret_cast<const f32x4*>(a.lnp_z + grow * 128 + q * 8 + hh * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      muv[it] = valid ? a.lnp_mean[grow] : 0.f;
      rsv[it] = valid ? a.lnp_rstd[grow] : 0.f;
    }
    FP_WLOAD(0, 0) FP_WLOAD(1, 1) FP_WLOAD(2, 2) FP_WLOAD(3, 3)      // first weight fragments: in flight beside the row arithmetic
    if constexpr (RING == 6) { FP_WLOAD(4, 4) FP_WLOAD(5, 5) }
#pragma unroll
    for (int it = 0; it < RP; ++it) {
      const int row = (tid >> 4) + it * 32, n = n0 - 2 + row;
      const bool inb = row < FP_HR && n >= 0 && n < a.N;
      const bool owned = inb && row >= 2 && row < 2 + FP_TOK;         // the 126 rows this workgroup writes and sums (halo rows: only the LDS tile)
      const bool ownvalid = owned && n < len_b;
      const size_t grow = (size_t)b * a.N + n;
      float dz[8];
      float s1 = 0.f, s2 = 0.f, g[8], xh[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = zv[it][e >> 2][e & 3];
        float d = dyv[it][e >> 2][e & 3];
        xh[e] = (z - muv[it]) * rsv[it];
        const float ln = xh[e] * wv[e] + bv[e];
        if (ownvalid) { gfg[e] += d * ln; gfb[e] += d; }
        d *= fg[e];
        if (ownvalid) { gw[e] += d * xh[e]; gb[e] += d; }
        g[e] = d * wv[e];
        s1 += g[e]; s2 += g[e] * xh[e];
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
      s1 *= (1.0f / 128); s2 *= (1.0f / 128);
#pragma unroll
      for (int e = 0; e < 8; ++e) dz[e] = rsv[it] * (g[e] - s1 - xh[e] * s2);
      if (owned) {
        if (dzs && row - 2 < DZ_ROWS) {
          *reinterpret_cast<f32x4*>(dzs + (row - 2) * 128 + q * 8) = f32x4{dz[0], dz[1], dz[2], dz[3]};
          *reinterpret_cast<f32x4*>(dzs + (row - 2) * 128 + q * 8 + 4) = f32x4{dz[4], dz[5], dz[6], dz[7]};
        } else {
          *reinterpret_cast<f32x4*>(a.Y + grow * a.ldy + q * 8) = f32x4{dz[0], dz[1], dz[2], dz[3]};
          *reinterpret_cast<f32x4*>(a.Y + grow * a.ldy + q * 8 + 4) = f32x4{dz[4], dz[5], dz[6], dz[7]};
        }
      }
      if (a.lnp_thresh) {
        float f[8];
        dx_dropout_scale4(seed, (unsigned long long)grow * 128 + q * 8, a.lnp_thresh, a.lnp_inv_keep, f);
        dx_dropout_scale4(seed, (unsigned long long)grow * 128 + q * 8 + 4, a.lnp_thresh, a.lnp_inv_keep, f + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) dz[e] *= f[e];
      }
      bf16x8 h8;
#pragma unroll
      for (int e = 0; e < 8; ++e) h8[e] = (dx_h16)dz[e];
      const f32x4 hv = __builtin_bit_cast(f32x4, h8);
      if (owned) *reinterpret_cast<f32x4*>(a.lnp_dg + grow * 128 + q * 8) = hv;
      if (row < FP_HR) *reinterpret_cast<f32x4*>(Xs + (q >> 3) * (FP_HR * 128) + fp_lds_off(row, q & 7)) = hv;
    }
    // affine / FiLM gradients: the four rows of a wave fold by shuffles, the eight waves through LDS (the first hidden image is still
    // unused), one atomic per channel and quantity per workgroup
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int off = 16; off < 64; off <<= 1) {
        gw[e] += __shfl_xor(gw[e], off, 64); gb[e] += __shfl_xor(gb[e], off, 64);
        gfg[e] += __shfl_xor(gfg[e], off, 64); gfb[e] += __shfl_xor(gfb[e], off, 64);
      }
    }
    float* const red = reinterpret_cast<float*>(Hs0);               // [8 waves][4 quantities][128]: 16 KB, below the image's zero rows
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wave * 4 + 0) * 128 + q * 8 + e] = gw[e]; red[(wave * 4 + 1) * 128 + q * 8 + e] = gb[e];
        red[(wave * 4 + 2) * 128 + q * 8 + e] = gfg[e]; red[(wave * 4 + 3) * 128 + q * 8 + e] = gfb[e];
      }
    }
    __syncthreads();
    {
      const int quant = tid >> 7, c = tid & 127;
      float t = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) t += red[(w8 * 4 + quant) * 128 + c];
      if (t != 0.f) {
        if (quant == 0) atomicAdd(&a.lnp_dw[c], t);
        else if (quant == 1) atomicAdd(&a.lnp_db[c], t);
        else if (a.lnp_dfilm) atomicAdd(&a.lnp_dfilm[(size_t)b * a.lnp_ld_dfilm + (quant == 3 ? 128 : 0) + c], t);
      }
    }
    __syncthreads();                                     // the scratch is read: the hidden image may be written
    if (tid < 64) {
      const int img = tid >> 5, rr = NROW + ((tid >> 4) & 1), qq = tid & 15;
      *reinterpret_cast<f32x4*>((img ? Hs1 : Hs0) + (qq >> 3) * (FP_HR * 128) + fp_lds_off(rr, qq & 7)) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  } else
  // ---- stage the activation tile: rows p = 0..129 <-> n = n0 - 2 + p, zero outside [0, NL) --------------------------------
  {
    f32x4 xr[XP];
#pragma unroll
    for (int it = 0; it < XP; ++it) {
      const int u = tid + it * 512;
      const int row = u >> 4, q = u & 15;
      const int n = n0 - 2 + row;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u < FP_HR * 16 && n >= 0 && n < NL) v = *reinterpret_cast<const f32x4*>(a.X + ((size_t)b * a.N + n) * a.ldx + q * 8);
      xr[it] = v;
    }
    FP_WLOAD(0, 0) FP_WLOAD(1, 1) FP_WLOAD(2, 2) FP_WLOAD(3, 3)      // first weight fragments: in flight beside the tile loads
    if constexpr (RING == 6) { FP_WLOAD(4, 4) FP_WLOAD(5, 5) }
    // rows 128, 129 of both hidden images stay zero for the whole kernel (the last two MFMA columns of the second conv read them)
    if (tid < 64) {
      const int img = tid >> 5, rr = NROW + ((tid >> 4) & 1), q = tid & 15;
      *reinterpret_cast<f32x4*>((img ? Hs1 : Hs0) + (q >> 3) * (FP_HR * 128) + fp_lds_off(rr, q & 7)) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < XP; ++it) {
      const int u = tid + it * 512;
      const int row = u >> 4, q = u & 15;
      if (u < FP_HR * 16) *reinterpret_cast<f32x4*>(Xs + (q >> 3) * (FP_HR * 128) + fp_lds_off(row, q & 7)) = xr[it];
    }
  }
  __syncthreads();
  FP_STAMP(2)

  // fragment read offsets inside an image: row = 16 j + r + tap, slot = (ks & 1) * 4 + g; (row & 7) does not depend on j, so the
  // swizzled offset is one value per (tap, ks & 1) plus compile-time constants for j and the channel chunk
  int foff[3][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) foff[t][hh] = fp_lds_off(r + t, hh * 4 + g);
  const int len_cols = min(FP_TOK, a.N - n0);         // output columns of this tile that exist
  const int h_rows = a.H ? len_cols : 0;              // rows of the mid activation to write out (none in a forward-only call: H == null)

  // Copy-out of the slice completed in the previous iteration (hidden rows 1..126 of the image <-> tokens n0 .. n0+125), as
  // full 128-byte row segments: piece K of 4 per thread, read from LDS in one matrix step and stored in the next (standalone it
  // stalled all eight waves for ~1 k cycles per iteration, stamped)
#define FP_CO_RD(K, IMGC)                                                                                            \
  {                                                                                                                  \
    const int u_ = fp_opaque(tid) + NROW * 8 + (K) * 256;       /* producers (threads 0..255): the upper half of the units */ \
    const int row_ = u_ >> 4, q_ = u_ & 15;                                                                          \
    cpv = *reinterpret_cast<const f32x4*>((IMGC) + (q_ >> 3) * (FP_HR * 128) + fp_lds_off(min(row_, NROW - 1) + 1, q_ & 7)); \
  }
#define FP_CO_ST(K, F0C)                                                                                             \
  {                                                                                                                  \
    const int u_ = fp_opaque(tid) + NROW * 8 + (K) * 256;                                                            \
    const int row_ = u_ >> 4, q_ = u_ & 15;                                                                          \
    if (row_ < h_rows) *reinterpret_cast<f32x4*>(a.H + ((size_t)b * a.N + n0 + row_) * a.ldh + (F0C) + q_ * 8) = cpv; \
  }
#define FP_COPY_OUT_CONS(IMGC, F0C)                                                                                  \
  {                                                                                                                  \
    f32x4 cv_[NJ / 2];                                                                                               \
    const int t_ = fp_opaque(tid) - 256;                       /* consumer threads are 256..511: the lower half of the units */ \
    _Pragma("unroll") for (int k = 0; k < NJ / 2; ++k) {                                                                  \
      const int u_ = t_ + k * 256;                                                                                   \
      cv_[k] = *reinterpret_cast<const f32x4*>((IMGC) + ((u_ & 15) >> 3) * (FP_HR * 128) + fp_lds_off((u_ >> 4) + 1, u_ & 7)); \
    }                                                                                                                \
    _Pragma("unroll") for (int k = 0; k < NJ / 2; ++k) {                                                             \
      const int u_ = t_ + k * 256;                                                                                   \
      const int row_ = u_ >> 4, q_ = u_ & 15;                                                                        \
      if (row_ < h_rows) *reinterpret_cast<f32x4*>(a.H + ((size_t)b * a.N + n0 + row_) * a.ldh + (F0C) + q_ * 8) = cv_[k]; \
    }                                                                                                                \
  }

  // one slice of one role: 12 steps of 16 MFMA; B fragments from `img` (the NEXT step's are read into the other register set
  // before this step's MFMAs), A fragments from the register ring (the slot just consumed is refilled four steps ahead, across
  // slice boundaries).  CO: copy-out pieces ride along on steps 0..7.  AUXPF: hook at step 8 (the backward's mask prefetch).
#define FP_RD8(DST, IMG, S)                                                                                          \
  {                                                                                                                  \
    const unsigned char* const bp_ = (IMG) + (((S) & 3) >> 1) * (FP_HR * 128) + foff[((S) >> 2) % 3][(S) & 1];       \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) DST[j] = *reinterpret_cast<const float4*>(bp_ + j * 2048);        \
  }
#define FP_STEP(CUR, NXT, IMG, GS0, S, AUXPF, FRESH, CO, IMGC, F0C)                                                  \
  {                                                                                                                  \
    if (DX_FP_ABL != 2 && DX_FP_ABL != 4) { if ((S) + 1 < 12) FP_RD8(NXT, IMG, (S) + 1) }                                              \
    else { _Pragma("unroll") for (int j = 0; j < NJ; ++j) NXT[j] = CUR[j]; }                                         \
    if ((CO) && DX_FP_ABL != 4 && (S) < NJ && ((S) & 1) == 0) FP_CO_RD((S) >> 1, IMGC)                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    const f32x4 w0 = wr[(S) % RING][0], w1 = wr[(S) % RING][1];                                                      \
    if ((FRESH) && (S) == 0) {                                  /* a producer slice starts from the bias as the C operand */ \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) { acc[0][j] = bv0; FP_MMA(w0, CUR[j], acc[0][j]) }              \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) { acc[1][j] = bv1; FP_MMA(w1, CUR[j], acc[1][j]) }              \
    } else {                                                                                                         \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) { FP_MMA(w0, CUR[j], acc[0][j]) }                               \
      _Pragma("unroll") for (int j = 0; j < NJ; ++j) { FP_MMA(w1, CUR[j], acc[1][j]) }                               \
    }                                                                                                                \
    if (DX_FP_ABL != 1 && DX_FP_ABL != 4) FP_WLOAD((S) % RING, (GS0) + (S) + RING)                                                     \
    if ((CO) && DX_FP_ABL != 4 && (S) < NJ && ((S) & 1) == 1) FP_CO_ST((S) >> 1, F0C)                                                  \
    if ((S) == 8) { AUXPF }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  }
#define FP_SLICE_STEPS(IMG, GS0, AUXPF, FR, CO, IMGC, F0C)                                                           \
  {                                                                                                                  \
    float4 xa[NJ], xb[NJ];                                                                                           \
    f32x4 cpv;                                                                                                       \
    FP_RD8(xa, IMG, 0)                                                                                               \
    FP_STEP(xa, xb, IMG, GS0, 0, AUXPF, FR, CO, IMGC, F0C) FP_STEP(xb, xa, IMG, GS0, 1, AUXPF, FR, CO, IMGC, F0C)    \
    FP_STEP(xa, xb, IMG, GS0, 2, AUXPF, FR, CO, IMGC, F0C) FP_STEP(xb, xa, IMG, GS0, 3, AUXPF, FR, CO, IMGC, F0C)    \
    FP_STEP(xa, xb, IMG, GS0, 4, AUXPF, FR, CO, IMGC, F0C) FP_STEP(xb, xa, IMG, GS0, 5, AUXPF, FR, CO, IMGC, F0C)    \
    FP_STEP(xa, xb, IMG, GS0, 6, AUXPF, FR, CO, IMGC, F0C) FP_STEP(xb, xa, IMG, GS0, 7, AUXPF, FR, CO, IMGC, F0C)    \
    FP_STEP(xa, xb, IMG, GS0, 8, AUXPF, FR, CO, IMGC, F0C) FP_STEP(xb, xa, IMG, GS0, 9, AUXPF, FR, CO, IMGC, F0C)    \
    FP_STEP(xa, xb, IMG, GS0, 10, AUXPF, FR, CO, IMGC, F0C) FP_STEP(xb, xa, IMG, GS0, 11, AUXPF, FR, CO, IMGC, F0C)  \
  }

  // The two roles run SEPARATE loops (same number of barriers in each): one loop with a role branch inside made the register
  // allocator carry the accumulators of both paths through common phis and rotate them through 64 extra registers.
  unsigned char* const stage = smem;                  // final 128 x 128 fp32 tile: 64 KB over the activation image and the first hidden image
  if (role == 0) {
    f32x4 acc[2][NJ];
    // hidden rows outside [0, N) are the second conv's zero padding: only the first / last tile of a batch row has any
    const bool edge = n0 == 0 || n0 + FP_TOK >= NL;
    const int tile_lin = b * tiles_n + n0 / FP_TOK;    // position of the tile in the padded grid (the same in forward and backward)
    // bias + ReLU (forward) or the sign mask (backward), bf16, into slice image f & 1
#define FP_PRODUCE(F_, COFLAG)                                                                                       \
    {                                                                                                                \
      const int f0 = slice_of(F_) << 7;                                                                              \
      /* the bias is the C operand of the slice's first MFMAs: no add in the epilogue */                             \
      const f32x4 bv0 = a.bias_a ? *reinterpret_cast<const f32x4*>(a.bias_a + f0 + 32 * wq + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};      \
      const f32x4 bv1 = a.bias_a ? *reinterpret_cast<const f32x4*>(a.bias_a + f0 + 32 * wq + 16 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f}; \
      bf16x4 av[2][MASK ? 1 : NJ];                                                                                   \
      unsigned mw[2] = {0u, 0u};                                                                                     \
      FP_SLICE_STEPS(Xs, (F_) * 12,                                                                                  \
        if constexpr (AUX) {                                                                                         \
          if constexpr (MASK) {                                                                                      \
            const unsigned* mp_ = a.hmask + ((((size_t)tile_lin * nslices + (f0 >> 7)) * 4 + wq) * 2) * 64 + fp_opaque(lane); \
            mw[0] = mp_[0]; mw[1] = mp_[64];                                                                         \
          } else {                                                                                                   \
          const int r2_ = fp_opaque(r);                                                                              \
          const int g2_ = fp_opaque(g);                                                                              \
          _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                         \
              const int n = min(max(n0 - 1 + 16 * j + r2_, 0), a.N - 1);                                             \
              av[i][j] = *reinterpret_cast<const bf16x4*>(a.aux + ((size_t)b * a.N + n) * a.ld_aux + f0 + 32 * wq + 16 * i + 4 * g2_); \
            }                                                                                                        \
          }                                                                                                          \
        }, true, COFLAG, ((((F_) - 1) & 1) ? Hs1 : Hs0), slice_of((F_) - 1) << 7)                                    \
      unsigned char* const out = ((F_) & 1) ? Hs1 : Hs0;                                                             \
      const int r_ = fp_opaque(r);                                                                                   \
      const int g_ = fp_opaque(g);                                                                                   \
      if (DX_FP_ABL == 3) { _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j) asm volatile("" :: "v"(acc[i][j])); } \
      else _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                           \
        const int c = 32 * wq + 16 * i + 4 * g_;                                                                     \
        unsigned char* const orow = out + (c >> 6) * (FP_HR * 128) + fp_lds_off(r_, (c & 63) >> 3) + ((g_ & 1) << 3); \
        unsigned sw_ = 0u;                                          /* forward: the sign word being built */        \
        const unsigned mwi_ = mw[i];                                                                                 \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                             \
          f32x4 v = acc[i][j];                                                                                       \
          if constexpr (AUX) {                                                                                       \
            if constexpr (MASK) {                                                                                    \
              /* bit -> all-ones / zero (v_bfe_i32), AND with the value's bits: two instructions per element instead of load + convert + compare + select */ \
              _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
                const float ve_ = v[e];                                                                              \
                const unsigned ke_ = __float_as_uint(ve_) & (unsigned)__builtin_amdgcn_sbfe((int)mwi_, NJ * 4 - 1 - (j * 4 + e), 1); \
                v[e] = __uint_as_float(ke_);                                                                         \
              }                                                                                                      \
            } else {                                                                                                 \
              _Pragma("unroll") for (int e = 0; e < 4; ++e) if (!((float)av[i][j][e] > 0.f)) v[e] = 0.f;             \
            }                                                                                                        \
          }                                                                                                          \
          if (edge) { const int n = n0 - 1 + 16 * j + r_; if (n < 0 || n >= NL) v = f32x4{0.f, 0.f, 0.f, 0.f}; }    \
          if constexpr (RELU) {                                                                                      \
            /* "> 0" of a float = its bits as a signed integer > 0: (bits - 1) has bit 31 clear; v_alignbit appends that bit to the word */ \
            if (a.hmask) {                                                                                           \
              _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
                const float ve_ = v[e];                                                                              \
                sw_ = __builtin_amdgcn_alignbit(sw_, __float_as_uint(ve_) - 1u, 31);                                 \
              }                                                                                                      \
            }                                                                                                        \
          }                                                                                                          \
          uint2 pk = fp_pack4(v[0], v[1], v[2], v[3]);                                                               \
          if constexpr (RELU) {                                                                                      \
            /* ReLU on the packed bf16 pairs: a negative float has its sign bit set, i.e. is a negative int16: max(., 0) zeroes it */ \
            pk.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk.x), s16x2{0, 0})); \
            pk.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk.y), s16x2{0, 0})); \
          }                                                                                                          \
          *reinterpret_cast<uint2*>(orow + j * 2048) = pk;                                                           \
        }                                                                                                            \
        if constexpr (RELU) {                                       /* one coalesced dword per lane: bit = 1 where the value was > 0 */ \
          if (a.hmask) a.hmask[((((size_t)tile_lin * nslices + (f0 >> 7)) * 4 + wq) * 2 + i) * 64 + fp_opaque(lane)] = ~sw_; \
        }                                                                                                            \
      }                                                                                                              \
    }
    FP_PRODUCE(0, false)
    __syncthreads();
    FP_STAMP(3)
    for (int it = 1; it < nslices; ++it) {
      if (it == 4) FP_STAMP(12)
      FP_PRODUCE(it, true)                            // + its half of the copy-out of slice it - 1, under the matrix steps
      if (it == 4) FP_STAMP(14)
      __syncthreads();
      FP_STAMP(3 + it)
    }
    {                                                 // last slice: the producers' half of its copy-out
      const unsigned char* const imgc = ((nslices - 1) & 1) ? Hs1 : Hs0;
      f32x4 cpv;
#pragma unroll
      for (int k = 0; k < NJ / 2; ++k) { FP_CO_RD(k, imgc) FP_CO_ST(k, slice_of(nslices - 1) << 7) }
    }
    __syncthreads();
    FP_STAMP(3 + nslices)
#undef FP_PRODUCE
  } else {
    f32x4 acc[2][NJ];
    const f32x4 bv0 = f32x4{0.f, 0.f, 0.f, 0.f}, bv1 = bv0;  // (names the step macro's producer-only branch refers to)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                   // iteration 0: the first slice is being produced
    FP_STAMP(3)
    for (int it = 1; it <= nslices; ++it) {
      if (it == 4) FP_STAMP(12)
      const int f = it - 1;                              // iteration whose slice is consumed; its hidden channels start at slice_of(f) * 128
      const unsigned char* const img = (f & 1) ? Hs1 : Hs0;
      // The consumers copy their half of the finished slice out BEFORE their matrix steps (4 pieces per thread; the producers'
      // half rides along on their matrix steps).  Both roles of a
      // SIMD advance through their MFMAs at the same rate, so starting together they also finish together and the producer's
      // epilogue ran with the matrix pipe idle; this head start for the producer (a "stagger", MI355X_MICROARCH.md "Two waves per
      // SIMD" item 9) puts its epilogue beside the consumer's last MFMAs instead.
      FP_COPY_OUT_CONS(img, slice_of(f) << 7)
      FP_SLICE_STEPS(img, f * 12, , false, false, img, 0)
      if (it == 4) FP_STAMP(13)
      __syncthreads();
      FP_STAMP(3 + it)
    }
    // every LDS read of the slice loop has retired (last barrier): the accumulators (+ bias) go to the staging tile, XOR-swizzled
    // 16-byte slots (conflict-free for these row-per-lane stores and for the row-major reads below)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int co = 32 * wq + 16 * i + 4 * g;
      const f32x4 bv = a.bias_b ? *reinterpret_cast<const f32x4*>(a.bias_b + co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int q = 16 * j + r;
        *reinterpret_cast<f32x4*>(stage + q * 512 + (((co >> 2) ^ (q & 15)) << 4)) = acc[i][j] + bv;
      }
    }
  }
#undef FP_SLICE_STEPS
#undef FP_STEP
#undef FP_RD8
#undef FP_COPY_OUT_CONS
#undef FP_CO_RD
#undef FP_CO_ST

  // ---- output: the consumers' 128 x 128 fp32 tile went through LDS and leaves as full 512-byte rows written by ALL eight waves
  //      (the four consumer waves alone, one 16-byte store per lane per (i, j), spent 7 k cycles on it while the producers idled)
  __syncthreads();
  if (a.ln_w) {
    // LayerNorm epilogue: a row of the tile = 32 consecutive lanes x 4 channels; statistics by 5 xor-shuffles inside the half wave
    const int s = tid & 31;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.ln_w + s * 4), bv = *reinterpret_cast<const f32x4*>(a.ln_b + s * 4);
    f32x4 fg = f32x4{1.f, 1.f, 1.f, 1.f}, fb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.film) {
      fg = *reinterpret_cast<const f32x4*>(a.film + (size_t)b * a.ld_film + s * 4);
      fb = *reinterpret_cast<const f32x4*>(a.film + (size_t)b * a.ld_film + 128 + s * 4);
    }
    const int len_b = a.lens ? a.lens[b] : a.N;
    const unsigned long long seed = a.seed_pre + (a.seed_offset ? *a.seed_offset : 0ull);
    bf16x8 qa[3][4];                                    // next block's in-projection (below): this wave's weight fragments, in flight during the row loop
    if (a.q_w) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qa[i][ks] = *reinterpret_cast<const bf16x8*>(a.q_w + (size_t)((wave * 3 + i) * 4 + ks) * 512 + lane * 8);
    }
    f32x4 resv[NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k) {                      // the residual rows of all passes are requested first
      const int row = (tid >> 5) + k * 16, n = n0 + row;
      resv[k] = (row < len_cols && n < len_b) ? *reinterpret_cast<const f32x4*>(a.ln_res + ((size_t)b * a.N + n) * 128 + s * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
      const int row = (tid >> 5) + k * 16, n = n0 + row;
      const bool inb = row < len_cols, valid = inb && n < len_b;
      const size_t grow = (size_t)b * a.N + n;
      f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
      if (valid) {
        z = *reinterpret_cast<const f32x4*>(stage + row * 512 + ((s ^ (row & 15)) << 4));
        if (a.thresh_pre) {
          float f[4];
          dx_dropout_scale4(seed, (unsigned long long)grow * 128 + s * 4, a.thresh_pre, a.inv_keep_pre, f);
          z[0] *= f[0]; z[1] *= f[1]; z[2] *= f[2]; z[3] *= f[3];
        }
        z += resv[k];
      }
      if (inb) *reinterpret_cast<f32x4*>(a.Y + grow * a.ldy + s * 4) = z;       // z, kept for the backward
      float sum = (z[0] + z[1]) + (z[2] + z[3]);
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) sum += __shfl_xor(sum, off, 64);
      const float mu = sum * (1.0f / 128);
      const f32x4 d = z - mu;
      float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) q += __shfl_xor(q, off, 64);
      const float rs = 1.0f / sqrtf(q * (1.0f / 128) + 1e-5f);
      if (inb && s == 0) { a.ln_mean[grow] = valid ? mu : 0.f; a.ln_rstd[grow] = valid ? rs : 0.f; }
      f32x4 y = d * rs * wv + bv;
      if (a.film) y = fg * y + fb;
      if (!valid) y = f32x4{0.f, 0.f, 0.f, 0.f};
      if (inb) *reinterpret_cast<f32x4*>(a.ln_y + grow * 128 + s * 4) = y;
      if (a.q_w) {
        // the same row as 16 bits, in place over the fp32 row it came from (this half wave has read all of it), for the GEMM below:
        // 16 slots of 8 channels, XOR-swizzled by the row; rows the tile does not own are zero operands
        bf16x4 h4;
        h4[0] = (dx_h16)y[0]; h4[1] = (dx_h16)y[1]; h4[2] = (dx_h16)y[2]; h4[3] = (dx_h16)y[3];
        *reinterpret_cast<bf16x4*>(stage + row * 512 + (((s >> 1) ^ (row & 15)) << 4) + ((s & 1) << 3)) = h4;
      }
    }
    if (a.q_w) {
      // QKV of the next block = y x W_in^T + bias.  A wave owns 48 of the 384 output channels for ALL rows of the tile: its 12 weight
      // fragments (A operands, straight from the pack, no two waves fetch the same one) were requested before the row loop above; the rows
      // (B operands) come from the 16-bit image just written.  96 MFMAs per wave.
      __syncthreads();
      f32x4 qb[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) qb[i] = *reinterpret_cast<const f32x4*>(a.q_bias + (wave * 3 + i) * 16 + g * 4);
#pragma unroll 2
      for (int j = 0; j < NJ; ++j) {
        const int row = j * 16 + r;
        bf16x8 xb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(stage + row * 512 + (((ks * 4 + g) ^ (row & 15)) << 4));
        f32x4 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = qb[i];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int i = 0; i < 3; ++i) acc[i] = DX_MFMA_H16(qa[i][ks], xb[ks], acc[i]);
        if (row < len_cols) {
          dx_h16* const orow = a.q_out + ((size_t)b * a.N + n0 + row) * 384 + wave * 48 + g * 4;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            bf16x4 o4;
            o4[0] = (dx_h16)acc[i][0]; o4[1] = (dx_h16)acc[i][1]; o4[2] = (dx_h16)acc[i][2]; o4[3] = (dx_h16)acc[i][3];
            *reinterpret_cast<bf16x4*>(orow + i * 16) = o4;
          }
        }
      }
    }
  } else if (a.lnb_w) {
    // LayerNorm-backward epilogue (the block's first LayerNorm): same arithmetic as ln_bwd_kernel<128, float, false>
    const int s = tid & 31;
    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.lnb_w + s * 4), bv = *reinterpret_cast<const f32x4*>(a.lnb_b + s * 4);
    (void)bv;
    const int len_b = a.lens ? a.lens[b] : a.N;
    const unsigned long long seed = a.seed_pre + (a.seed_offset ? *a.seed_offset : 0ull);
    f32x4 oldv[NJ], zv[NJ];
    float muv[NJ], rsv[NJ];
#pragma unroll
    for (int k = 0; k < NJ; ++k) {                      // all global reads of the passes first
      const int row = (tid >> 5) + k * 16, n = n0 + row;
      const bool valid = row < len_cols && n < len_b;
      const size_t grow = (size_t)b * a.N + n;
      if (dzs && row < DZ_ROWS) oldv[k] = valid ? *reinterpret_cast<const f32x4*>(dzs + row * 128 + s * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      else oldv[k] = valid ? *reinterpret_cast<const f32x4*>(a.Y + grow * a.ldy + s * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      zv[k] = valid ? *reinterpret_cast<const f32x4*>(a.lnb_z + grow * 128 + s * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      muv[k] = valid ? a.lnb_mean[grow] : 0.f;
      rsv[k] = valid ? a.lnb_rstd[grow] : 0.f;
    }
    f32x4 gw = f32x4{0.f, 0.f, 0.f, 0.f}, gb = gw;
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
      const int row = (tid >> 5) + k * 16, n = n0 + row;
      const bool inb = row < len_cols, valid = inb && n < len_b;
      const size_t grow = (size_t)b * a.N + n;
      f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
      if (valid) d = *reinterpret_cast<const f32x4*>(stage + row * 512 + ((s ^ (row & 15)) << 4)) + oldv[k];
      const f32x4 xh = (zv[k] - muv[k]) * rsv[k];
      gw += d * xh; gb += d;
      const f32x4 g = d * wv;
      float s1 = (g[0] + g[1]) + (g[2] + g[3]);
      const f32x4 gx = g * xh;
      float s2 = (gx[0] + gx[1]) + (gx[2] + gx[3]);
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
      s1 *= (1.0f / 128); s2 *= (1.0f / 128);
      f32x4 dz = (g - s1 - xh * s2) * rsv[k];
      if (!inb) continue;
      *reinterpret_cast<f32x4*>(a.Y + grow * a.ldy + s * 4) = dz;                  // dz1: the residual-branch gradient (the dz2 row it replaces is dead)
      if (a.thresh_pre) {
        float f[4];
        dx_dropout_scale4(seed, (unsigned long long)grow * 128 + s * 4, a.thresh_pre, a.inv_keep_pre, f);
        dz[0] *= f[0]; dz[1] *= f[1]; dz[2] *= f[2]; dz[3] *= f[3];
      }
      bf16x4 h4;
      h4[0] = (dx_h16)dz[0]; h4[1] = (dx_h16)dz[1]; h4[2] = (dx_h16)dz[2]; h4[3] = (dx_h16)dz[3];
      *reinterpret_cast<bf16x4*>(a.lnb_dg + grow * 128 + s * 4) = h4;
      // the same 16-bit row, in place over the fp32 row it came from (this half wave has read all of it), for the GEMM below:
      // 16 slots of 8 channels, XOR-swizzled by the row
      if (a.lnb_wt) *reinterpret_cast<bf16x4*>(stage + row * 512 + (((s >> 1) ^ (row & 15)) << 4) + ((s & 1) << 3)) = h4;
    }
    if (a.lnb_wt) {                                     // rows this tile does not own (beyond N): zero operands
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        const int row = (tid >> 5) + k * 16;
        if (row >= len_cols) *reinterpret_cast<uint2*>(stage + row * 512 + (((s >> 1) ^ (row & 15)) << 4) + ((s & 1) << 3)) = make_uint2(0u, 0u);
      }
    }
    // affine gradients: 16 row groups x 128 channels fold through LDS (behind the staging tile), one atomic per channel per workgroup
    float* const red = reinterpret_cast<float*>(smem + NROW * 512);
    *reinterpret_cast<f32x4*>(red + (tid >> 5) * 128 + s * 4) = gw;
    *reinterpret_cast<f32x4*>(red + 2048 + (tid >> 5) * 128 + s * 4) = gb;
    __syncthreads();
    if (tid < 256) {
      const float* src = red + (tid >> 7) * 2048 + (tid & 127);
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += src[q * 128];
      if (t != 0.f) atomicAdd((tid >> 7) ? &a.lnb_db[tid & 127] : &a.lnb_dw[tid & 127], t);
    }
    if (a.lnb_wt) {
      // DATT = dg1 x W_out: a wave takes 16 rows of the tile (B operand from the 16-bit rows above) and, with NJ < 8 row blocks, a share of
      // the 128 output channels of the transposed weight (A operand fragments straight from the pack); (the barrier above ordered the row writes)
      constexpr int CB = NJ;                             // 16-channel blocks per wave: 8 waves cover NJ row blocks x 8 channel blocks
      const int rb = wave % NJ, cb0 = (wave / NJ) * CB;
      const int row = rb * 16 + r;
      bf16x8 xb[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(stage + row * 512 + (((ks * 4 + g) ^ (row & 15)) << 4));
      f32x4 acc[CB];
#pragma unroll
      for (int i = 0; i < CB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 wa[CB];
#pragma unroll
        for (int i = 0; i < CB; ++i) wa[i] = *reinterpret_cast<const bf16x8*>(a.lnb_wt + (size_t)((cb0 + i) * 4 + ks) * 512 + lane * 8);
#pragma unroll
        for (int i = 0; i < CB; ++i) acc[i] = DX_MFMA_H16(wa[i], xb[ks], acc[i]);
      }
      if (row < len_cols) {
        dx_h16* const orow = a.lnb_datt + ((size_t)b * a.N + n0 + row) * 128 + cb0 * 16 + g * 4;
#pragma unroll
        for (int i = 0; i < CB; ++i) {
          bf16x4 o4;
          o4[0] = (dx_h16)acc[i][0]; o4[1] = (dx_h16)acc[i][1]; o4[2] = (dx_h16)acc[i][2]; o4[3] = (dx_h16)acc[i][3];
          *reinterpret_cast<bf16x4*>(orow + i * 16) = o4;
        }
      }
    }
  } else {
    f32x4 old[NJ];
    if (a.accumulate) {
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        const int u = tid + k * 512;
        const int row = min(u >> 5, len_cols - 1), s = u & 31;
        old[k] = *reinterpret_cast<const f32x4*>(a.Y + ((size_t)b * a.N + n0 + row) * a.ldy + s * 4);
      }
    }
#pragma unroll
    for (int k = 0; k < NJ; ++k) {
      const int u = tid + k * 512;
      const int row = u >> 5, s = u & 31;
      f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 512 + ((s ^ (row & 15)) << 4));
      if (a.accumulate) v += old[k];
      if (row < len_cols) *reinterpret_cast<f32x4*>(a.Y + ((size_t)b * a.N + n0 + row) * a.ldy + s * 4) = v;
    }
  }
  FP_STAMP(15)
#undef FP_WLOAD
}

#undef FP_MMA

}  // namespace

#ifdef DX_FFPAIR_STAMPS
static unsigned long long* g_ffpair_stamps = nullptr;
extern "C" void dx_ff_pair_set_stamps(unsigned long long* p) { g_ffpair_stamps = p; }
#endif

extern "C" {

// One launch for conv(k=3, 128 -> F) -> mid -> conv(k=3, F -> 128) on channels-last bf16 activations (see the header of this file).
// Wa / Wb: fragment-major bf16 packs written by dx_pack_weights (forward pair: conv1.fwd / conv2.fwd; input-gradient pair:
// conv2.bwd / conv1.bwd).  H [B][N][F] bf16 receives the mid activation; Y [B][N][128] fp32 the result (+= if accumulate).
// relu_mid: mid = ReLU.  aux (optional, bf16 [B][N][F]): mid zeroes every position where aux <= 0 (exactly one of the two).
// skip_halo: token tiles that start at or beyond min(lens[b] + skip_halo, N) are padding nobody reads: zero-filled, not computed.
static int ff_pair_launch(const void* X, int ldx, const void* Wa, const void* Wb, const float* bias_a, const float* bias_b,
                          const void* aux, int ld_aux, void* H, int ldh, float* Y, int ldy,
                          int B, int N, int F, int relu_mid, int accumulate, const int* lens, int skip_halo, const int* rows_exist,
                          const float* ln_res, const float* ln_w, const float* ln_b, const float* film, int ld_film,
                          float* ln_y, float* ln_mean, float* ln_rstd, unsigned long long seed_pre, float p_pre, const unsigned long long* seed_offset,
                          const float* lnb_z, const float* lnb_mean, const float* lnb_rstd, const float* lnb_w, const float* lnb_b,
                          void* lnb_dg, float* lnb_dw, float* lnb_db, const FFPairArgs* prologue, void* stream, unsigned* hmask = nullptr) {
  DX_REQUIRE((X || prologue) && Wa && Wb && Y, "dx_ff_pair: null pointer");
  DX_REQUIRE(H || (relu_mid && !lnb_w), "dx_ff_pair: only the forward pair may omit the mid-activation output H");
  if (ln_w) {
    DX_REQUIRE(relu_mid && !accumulate && lens && skip_halo >= 0 && ldy == 128, "dx_ff_pair_ln: forward pair, no accumulate, lens, dense Z");
    DX_REQUIRE(ln_res && ln_b && ln_y && ln_mean && ln_rstd, "dx_ff_pair_ln: null pointer");
    DX_REQUIRE(!film || ld_film >= 256, "dx_ff_pair_ln: ld_film too small");
    DX_REQUIRE(p_pre >= 0.f && p_pre < 1.f, "dx_ff_pair_ln: dropout p out of range");
  }
  if (lnb_w) {
    DX_REQUIRE(aux && accumulate && lens && skip_halo >= 0 && ldy == 128 && !ln_w, "dx_ff_pair_lnbwd: input-gradient pair accumulating onto a dense Y, lens required");
    DX_REQUIRE(lnb_z && lnb_mean && lnb_rstd && lnb_b && lnb_dg && lnb_dw && lnb_db, "dx_ff_pair_lnbwd: null pointer");
    DX_REQUIRE(p_pre >= 0.f && p_pre < 1.f && ((uintptr_t)lnb_dg % 16) == 0, "dx_ff_pair_lnbwd: bad dropout p or alignment");
  }
  DX_REQUIRE(B > 0 && N > 0 && F >= 128 && (F % 128) == 0, "dx_ff_pair: bad dims B=%d N=%d F=%d (F must be a multiple of 128)", B, N, F);
  DX_REQUIRE(ldx >= 128 && (ldx % 8) == 0 && ldh >= F && (ldh % 8) == 0 && ldy >= 128 && (ldy % 4) == 0, "dx_ff_pair: bad leading dimensions");
  DX_REQUIRE(!aux || (ld_aux >= F && (ld_aux % 4) == 0), "dx_ff_pair: bad ld_aux");
  DX_REQUIRE((aux != nullptr) != (relu_mid != 0), "dx_ff_pair: exactly one of relu_mid (forward) and aux (backward) must be given");
  DX_REQUIRE(skip_halo < 0 || lens, "dx_ff_pair: skip_halo needs lens");
  if (prologue && prologue->q_w) {
    DX_REQUIRE(ln_w && prologue->q_bias && prologue->q_out && ((uintptr_t)prologue->q_w % 16) == 0 && ((uintptr_t)prologue->q_out % 16) == 0,
               "dx_ff_pair_ln_qkv: needs the LayerNorm epilogue, a bias and 16-byte aligned pointers");
  } else if (prologue) {
    DX_REQUIRE(lnb_w && prologue->lnp_dy && prologue->lnp_z && prologue->lnp_mean && prologue->lnp_rstd && prologue->lnp_w && prologue->lnp_b &&
               prologue->lnp_dg && prologue->lnp_dw && prologue->lnp_db, "dx_ff_block_bwd: null pointer");
    DX_REQUIRE((prologue->lnp_film == nullptr) == (prologue->lnp_dfilm == nullptr), "dx_ff_block_bwd: film and dfilm must come together");
    DX_REQUIRE(((uintptr_t)prologue->lnp_dy % 16) == 0 && ((uintptr_t)prologue->lnp_z % 16) == 0 && ((uintptr_t)prologue->lnp_dg % 16) == 0,
               "dx_ff_block_bwd: pointers must be 16-byte aligned");
  }
  DX_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)Wa % 16) == 0 && ((uintptr_t)Wb % 16) == 0 && ((uintptr_t)H % 16) == 0 &&
             ((uintptr_t)Y % 16) == 0 && ((uintptr_t)aux % 8) == 0, "dx_ff_pair: pointers must be 16-byte aligned");
  FFPairArgs a{(const dx_h16*)X, ldx, (const dx_h16*)Wa, (const dx_h16*)Wb, bias_a, bias_b, (const dx_h16*)aux, ld_aux,
               (dx_h16*)H, ldh, Y, ldy, B, N, F, accumulate, lens, skip_halo, rows_exist,
               ln_res, ln_w, ln_b, film, ld_film, ln_y, ln_mean, ln_rstd, seed_pre, (unsigned)lrintf(p_pre * 65536.f), 1.f / (1.f - p_pre), seed_offset,
               nullptr, nullptr, nullptr,
               lnb_z, lnb_mean, lnb_rstd, lnb_w, lnb_b, (dx_h16*)lnb_dg, lnb_dw, lnb_db};
  if (prologue && prologue->q_w) { a.q_w = prologue->q_w; a.q_bias = prologue->q_bias; a.q_out = prologue->q_out; }
  if (prologue && prologue->lnp_dy) {
    a.lnp_dy = prologue->lnp_dy; a.lnp_z = prologue->lnp_z; a.lnp_mean = prologue->lnp_mean; a.lnp_rstd = prologue->lnp_rstd;
    a.lnp_w = prologue->lnp_w; a.lnp_b = prologue->lnp_b; a.lnp_film = prologue->lnp_film; a.lnp_ld_film = prologue->lnp_ld_film;
    a.lnp_dg = prologue->lnp_dg; a.lnp_dw = prologue->lnp_dw; a.lnp_db = prologue->lnp_db; a.lnp_dfilm = prologue->lnp_dfilm;
    a.lnp_ld_dfilm = prologue->lnp_ld_dfilm; a.lnp_seed = prologue->lnp_seed; a.lnp_thresh = prologue->lnp_thresh; a.lnp_inv_keep = prologue->lnp_inv_keep;
    a.lnb_wt = prologue->lnb_wt; a.lnb_datt = prologue->lnb_datt;
  }
#ifdef DX_FFPAIR_STAMPS
  a.stamps = g_ffpair_stamps;
#endif
  // LDS: the three images, + (block backward with the LayerNorm prologue) the dz2 rows that wait for the epilogue
  constexpr int DZ8 = 112 * 512, DZ4 = FP<4>::TOK * 512;
  static const int dz_env = getenv("DX_FF_DZ_LDS") ? atoi(getenv("DX_FF_DZ_LDS")) : 1;
  a.dz_lds = dz_env && a.lnp_w != nullptr;
  const int lds8 = 3 * FP<8>::IMG + (a.dz_lds ? DZ8 : 0), lds4 = 3 * FP<4>::IMG + (a.dz_lds ? DZ4 : 0);
  static bool configured = false;
  if (!configured) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_pair_kernel<true, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * FP<8>::IMG + DZ8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_pair_kernel<true, false, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * FP<8>::IMG + DZ8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_pair_kernel<true, false, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * FP<4>::IMG + DZ4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_pair_kernel<false, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * FP<8>::IMG);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_pair_kernel<true, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * FP<4>::IMG + DZ4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_pair_kernel<false, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * FP<4>::IMG);
    configured = true;
  }
  // tile width: 62-token tiles with two workgroups per CU (NJ = 4) or 126-token tiles with one (NJ = 8); DX_FF_NJ overrides (diagnostics)
  a.hmask = hmask;
  static const int skew_env = getenv("DX_FF_SKEW") ? atoi(getenv("DX_FF_SKEW")) : 0;     // measured neutral (56.5 vs 56.6 us): off
  a.slice_skew = skew_env;
  static const int nj_env = getenv("DX_FF_NJ") ? atoi(getenv("DX_FF_NJ")) : 0;
  // Tile width by shape: 126-token tiles where they fill the chip (frame axis: ~250 live tiles at C2); 62-token tiles for short batches
  // (symbol axis: 48 utterances of <= 120 symbols are 48 tiles of 126 but 96 of 62: twice the CUs, half the serial slice loop per workgroup).
  // Forward and backward see the same (B, N) and therefore make the same choice (the sign words are laid out per tile).
  const int nj = nj_env == 8 || nj_env == 4 ? nj_env : (B * dx_cdiv(N, FP<8>::TOK) >= 96 ? 8 : 4);
  hipStream_t s = (hipStream_t)stream;
  dx_prof_begin(DX_PROF_CONV_GEMM, s);
  if (nj == 8) {
    if (aux && hmask) hipLaunchKernelGGL((ff_pair_kernel<true, false, 8, true>), dim3(B * dx_cdiv(N, FP<8>::TOK)), dim3(512), lds8, s, a);
    else if (aux) hipLaunchKernelGGL((ff_pair_kernel<true, false, 8>), dim3(B * dx_cdiv(N, FP<8>::TOK)), dim3(512), lds8, s, a);
    else hipLaunchKernelGGL((ff_pair_kernel<false, true, 8>), dim3(B * dx_cdiv(N, FP<8>::TOK)), dim3(512), 3 * FP<8>::IMG, s, a);
  } else {
    if (aux && hmask) hipLaunchKernelGGL((ff_pair_kernel<true, false, 4, true>), dim3(B * dx_cdiv(N, FP<4>::TOK)), dim3(512), lds4, s, a);
    else if (aux) hipLaunchKernelGGL((ff_pair_kernel<true, false, 4>), dim3(B * dx_cdiv(N, FP<4>::TOK)), dim3(512), lds4, s, a);
    else hipLaunchKernelGGL((ff_pair_kernel<false, true, 4>), dim3(B * dx_cdiv(N, FP<4>::TOK)), dim3(512), 3 * FP<4>::IMG, s, a);
  }
  dx_prof_end(DX_PROF_CONV_GEMM, s);
  DX_LAUNCH_CHECK("dx_ff_pair");
  return DX_OK;
}

int dx_ff_pair(const void* X, int ldx, const void* Wa, const void* Wb, const float* bias_a, const float* bias_b,
               const void* aux, int ld_aux, void* H, int ldh, float* Y, int ldy,
               int B, int N, int F, int relu_mid, int accumulate, const int* lens, int skip_halo, const int* rows_exist, void* stream) {
  return ff_pair_launch(X, ldx, Wa, Wb, bias_a, bias_b, aux, ld_aux, H, ldh, Y, ldy, B, N, F, relu_mid, accumulate, lens, skip_halo, rows_exist,
                        nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0ull, 0.f, nullptr,
                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

// The forward pair with the block's second LayerNorm folded into its epilogue:
//   Z = res + dropout(conv3(Wb, ReLU(conv3(Wa, X) + bias_a)) + bias_b)   (fp32 [B][N][128], kept for the backward)
//   Yln = mask(FiLM(LayerNorm(Z)))                                         (the arguments of dx_ln_fwd with C = 128, halo 0)
int dx_ff_pair_ln(const void* X, int ldx, const void* Wa, const void* Wb, const float* bias_a, const float* bias_b, void* H, int ldh, float* Z,
                  int B, int N, int F, const int* lens, int skip_halo, const int* rows_exist,
                  const float* res, const float* ln_w, const float* ln_b, const float* film, int ld_film, float* Yln, float* mean, float* rstd,
                  uint64_t seed_pre, float p_pre, const uint64_t* seed_offset, void* hmask, void* stream) {
  DX_REQUIRE(ln_w != nullptr, "dx_ff_pair_ln: null pointer");
  return ff_pair_launch(X, ldx, Wa, Wb, bias_a, bias_b, nullptr, 0, H, ldh, Z, 128, B, N, F, 1, 0, lens, skip_halo, rows_exist,
                        res, ln_w, ln_b, film, ld_film, Yln, mean, rstd, (unsigned long long)seed_pre, p_pre, (const unsigned long long*)seed_offset,
                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream, (unsigned*)hmask);
}

// dx_ff_pair_ln + the NEXT block's attention in-projection on the normalised tile: QKV = Yln x Wq^T + bias_q (16-bit [B][N][384]; Wq: the
// forward pack of the (384, 128) in-projection weight) -- the result of dx_conv_gemm(Yln, Wq, bias_q, taps = 1, lens, halo 0) without its launch
int dx_ff_pair_ln_qkv(const void* X, int ldx, const void* Wa, const void* Wb, const float* bias_a, const float* bias_b, void* H, int ldh, float* Z,
                      int B, int N, int F, const int* lens, int skip_halo, const int* rows_exist,
                      const float* res, const float* ln_w, const float* ln_b, const float* film, int ld_film, float* Yln, float* mean, float* rstd,
                      uint64_t seed_pre, float p_pre, const uint64_t* seed_offset, const void* Wq, const float* bias_q, void* QKV, void* hmask, void* stream) {
  DX_REQUIRE(ln_w != nullptr && Wq != nullptr, "dx_ff_pair_ln_qkv: null pointer");
  FFPairArgs ext{};
  ext.q_w = (const dx_h16*)Wq; ext.q_bias = bias_q; ext.q_out = (dx_h16*)QKV;
  return ff_pair_launch(X, ldx, Wa, Wb, bias_a, bias_b, nullptr, 0, H, ldh, Z, 128, B, N, F, 1, 0, lens, skip_halo, rows_exist,
                        res, ln_w, ln_b, film, ld_film, Yln, mean, rstd, (unsigned long long)seed_pre, p_pre, (const unsigned long long*)seed_offset,
                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ext, stream, (unsigned*)hmask);
}

// The input-gradient pair with the backward of the block's FIRST LayerNorm folded into its epilogue:
//   dy1 = Y (the residual-branch gradient dz2, already there) + conv1^T(mask(conv2^T(X)))          as dx_ff_pair(accumulate = 1)
//   Y <- dz1 = LayerNorm-backward(dy1; z, mean, rstd, w)   DG <- dropout(dz1) (16-bit)   dw / db += the affine gradients
// i.e. dx_ln_bwd(C = 128, no FiLM, halo 0, 16-bit shadow) on the output tile while it is in LDS.
int dx_ff_pair_lnbwd(const void* X, int ldx, const void* Wa, const void* Wb, const void* aux, int ld_aux, void* H, int ldh, float* Y,
                     int B, int N, int F, const int* lens, int skip_halo,
                     const float* z, const float* mean, const float* rstd, const float* ln_w, const float* ln_b, void* DG, float* dw, float* db,
                     uint64_t seed_pre, float p_pre, const uint64_t* seed_offset, void* stream) {
  DX_REQUIRE(ln_w != nullptr, "dx_ff_pair_lnbwd: null pointer");
  return ff_pair_launch(X, ldx, Wa, Wb, nullptr, nullptr, aux, ld_aux, H, ldh, Y, 128, B, N, F, 0, 1, lens, skip_halo, nullptr,
                        nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, (unsigned long long)seed_pre, p_pre, (const unsigned long long*)seed_offset,
                        z, mean, rstd, ln_w, ln_b, DG, dw, db, nullptr, stream);
}

// The whole conv feed-forward half of an FFT block's backward in ONE launch:
//   [prologue] dz2 = LayerNorm2-backward(dY2; z2, mean2, rstd2, w2, film)  -> Y;  dropout(dz2) (16 bits) -> the LDS tile and DG2; dw2 / db2 / dfilm +=
//   [pair]     conv2^T -> ReLU mask (aux) -> conv1^T, hidden gradient -> H
//   [epilogue] dz1 = LayerNorm1-backward(Y + pair result; z1, mean1, rstd1, w1) -> Y;  dropout(dz1) (16 bits) -> DG1;  dw1 / db1 +=
// i.e. dx_ln_bwd (C = 128, FiLM optional, halo 0) + dx_ff_pair (input-gradient pair) + dx_ln_bwd (no FiLM) of model.py:225-233 / :206-217 /
// :188-191 backward.  Y is an OUTPUT here (nothing is read from it); dfilm (optional, [B][ld_dfilm >= 256]) accumulates like dw / db.
int dx_ff_block_bwd(const float* dY2, const float* z2, const float* mean2, const float* rstd2, const float* ln2_w, const float* ln2_b,
                    const float* film, int ld_film, void* DG2, float* dw2, float* db2, float* dfilm, int ld_dfilm, uint64_t seed2, float p2,
                    const void* Wa, const void* Wb, const void* aux, int ld_aux, void* H, int ldh, float* Y,
                    int B, int N, int F, const int* lens, int skip_halo,
                    const float* z1, const float* mean1, const float* rstd1, const float* ln1_w, const float* ln1_b, void* DG1, float* dw1, float* db1,
                    uint64_t seed1, float p1, const void* Wout_bwd, void* DATT, const uint64_t* seed_offset, const void* hmask, void* stream) {
  DX_REQUIRE(ln1_w != nullptr && ln2_w != nullptr, "dx_ff_block_bwd: null pointer");
  DX_REQUIRE((Wout_bwd == nullptr) == (DATT == nullptr) && ((uintptr_t)Wout_bwd % 16) == 0 && ((uintptr_t)DATT % 16) == 0, "dx_ff_block_bwd: Wout_bwd and DATT come together, 16-byte aligned");
  DX_REQUIRE(p2 >= 0.f && p2 < 1.f, "dx_ff_block_bwd: dropout p out of range");
  DX_REQUIRE(!film || (ld_film >= 256 && ld_dfilm >= 256), "dx_ff_block_bwd: ld_film / ld_dfilm too small");
  FFPairArgs pro{};
  pro.lnp_dy = dY2; pro.lnp_z = z2; pro.lnp_mean = mean2; pro.lnp_rstd = rstd2; pro.lnp_w = ln2_w; pro.lnp_b = ln2_b;
  pro.lnp_film = film; pro.lnp_ld_film = ld_film; pro.lnp_dg = (dx_h16*)DG2; pro.lnp_dw = dw2; pro.lnp_db = db2; pro.lnp_dfilm = dfilm;
  pro.lnb_wt = (const dx_h16*)Wout_bwd; pro.lnb_datt = (dx_h16*)DATT;
  pro.lnp_ld_dfilm = ld_dfilm; pro.lnp_seed = (unsigned long long)seed2; pro.lnp_thresh = (unsigned)lrintf(p2 * 65536.f); pro.lnp_inv_keep = 1.f / (1.f - p2);
  return ff_pair_launch(nullptr, 128, Wa, Wb, nullptr, nullptr, aux, ld_aux, H, ldh, Y, 128, B, N, F, 0, 1, lens, skip_halo, nullptr,
                        nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, (unsigned long long)seed1, p1, (const unsigned long long*)seed_offset,
                        z1, mean1, rstd1, ln1_w, ln1_b, DG1, dw1, db1, &pro, stream, (unsigned*)hmask);
}

}  // extern "C"

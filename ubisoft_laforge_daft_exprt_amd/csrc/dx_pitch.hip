// The frozen pitch predictor of the pitch-consistency loss term, one launch per direction (16-bit MFMA operands, gfx950).
//
//   reference: layers/pitch_predictor.py:38-74 (three weight-normed Conv1d(k = 3) + ReLU + BatchNorm1d(eval) stages, 80 -> 256 -> 256 -> 256,
//   and a Conv1d(k = 3) to one channel), applied to the PREDICTED mel in loss.py:131-140; only the gradient with respect to the mel is needed.
//
// As separate launches (dx_conv_gemm per layer + dx_channel_affine + two transposes) the chain was ~15 launches and ~0.3 ms of the C2 step for
// 26 GFLOP each way: 256-channel layers are four-stage K loops, i.e. latency shells, and every layer's 512 B/token activation went through HBM
// twice.  Here a workgroup owns 122 output tokens: the activations of its 130-row window live in two LDS images (256 channels x 132 rows, bf16,
// 64-channel chunks of 128-byte rows with the XOR swizzle of the conv kernels), every layer is one MFMA pass image -> image (eight waves, each 32
// output channels x 128 rows, weight fragments straight from the fragment-major packs into a register ring), and HBM sees the mel window, the
// prediction and 96 bytes per token of ReLU sign bits (the backward needs the masks, not the activations: the network is frozen).
// A layer computes 128 rows starting one row further in than the previous one (rows l + 1 .. l + 128 of the window), so the rows that miss a
// neighbour at the window's ends are garbage that never reaches a valid row: 4 + 122 + 4 valid input rows -> 122 valid outputs.
//   forward:   mel (B, 80, T) fp32 -> window image -> L0 -> L1 -> L2 (MFMA, epilogue: bias, ReLU, sign bits, BatchNorm affine, zero outside
//              [0, T)) -> L3 (one output channel: VALU dot products) -> pp (B, T) fp32; masks (B, T, 3, 8) uint32
//   backward:  dpp (B, T) -> conv^T(W3) (VALU) x scale2 x mask2 -> conv^T(W2) x scale1 x mask1 -> conv^T(W1) x scale0 x mask0 -> conv^T(W0)
//              -> dmel (B, 80, T) += (MFMA passes use the transposed + flipped packs dx_pack_weights already writes)
#include "dx_common.h"
#include <algorithm>

namespace {

constexpr int PC_TOK = 122;            // output tokens per tile
constexpr int PC_IR = 132;             // rows of an LDS image: row p <-> token n0 - 4 + p
constexpr int PC_CHUNK = PC_IR * 128;  // bytes of one 64-channel chunk image
constexpr int PC_IMG = 4 * PC_CHUNK;   // 256 channels
constexpr int PC_MASK = 3 * PC_IR * 32;   // bytes of the sign-bit staging area: [layer][row][8 dwords]
constexpr int PC_PRE = 4096, PC_RED = 2048, PC_W3 = 3072;   // scratch: tile prefix [B + 1]; partial sums of the last layer / the window of dpp; the last conv's weights
constexpr int PC_SMEM = 2 * PC_IMG + PC_MASK + PC_PRE + PC_RED + PC_W3;
constexpr int PC_MAX_B = 1000;

__device__ __forceinline__ int pc_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }

struct PitchArgs {
  const float* mel; float* dmel; int B, M, T;          // (B, M, T) fp32, M = 80
  const int* lens;
  const dx_h16* w0; const dx_h16* w1; const dx_h16* w2;   // forward: the forward packs of layers 0..2; backward: the backward packs of layers 0..2
  const float* b0; const float* b1; const float* b2;
  const float* s0; const float* s1; const float* s2;   // BatchNorm scale (eval mode, folded)
  const float* t0; const float* t1; const float* t2;   // BatchNorm shift
  const float* w3; float b3;                           // the last conv's row 0: (256, 3) fp32 as in the checkpoint layout (C, taps)
  float* pp; const float* dpp;                         // (B, T)
  unsigned* masks;                                     // (B, T, 3, 8): bit c of layer l at token n = "pre-activation of channel c was > 0"
};

#define PC_MMA(W, X, C) C = DX_MFMA_H16(__builtin_bit_cast(bf16x8, W), __builtin_bit_cast(bf16x8, X), C);

// One MFMA pass image -> accumulators: out[32 channels of this wave (NB row blocks of 16)][128 rows] = sum over taps and K of
// pack[tap][row block][k] x in[row - 1 + tap][k].  pack: fragment-major (dx_gemm.hip wb_off), rows16 = padded rows / 16, kblocks = padded K / 32;
// NKS = K steps of 32 channels actually walked per tap; rb0 = this wave's first row block; base = first output row of the window.
template <int NKS, int NB>
__device__ __forceinline__ void pc_mfma_pass(const dx_h16* __restrict__ pack, int rows16, int kblocks, int rb0, const unsigned char* in, int base,
                                             int lane, f32x4 (&acc)[NB][8]) {
  constexpr int STEPS = 3 * NKS, RING = 4;
  const int r = lane & 15, g = lane >> 4;
  f32x4 wr[RING][NB];
  // Fragment addresses = a per-lane base + immediates; loop-invariant across tiles, hipcc hoists a dozen 64-bit bases per pass out of the tile
  // loop and spills them (a scratch reload in the middle of a pass drains the weight ring: vmcnt is in order).  The asm keeps them per pass.
  unsigned lane16 = lane * 16;
  asm volatile("" : "+v"(lane16));
  auto wload = [&](int slot, int s) {
    const int tap = s / NKS, ks = s - tap * NKS;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const char* sb = reinterpret_cast<const char*>(pack + ((size_t)(tap * rows16 + rb0 + i) * kblocks + ks) * 512);
      wr[slot][i] = *reinterpret_cast<const f32x4*>(sb + lane16);
    }
  };
  auto xread = [&](float4 (&x)[8], int s) {
    const int tap = s / NKS, ks = s - tap * NKS;
    const unsigned char* p = in + (ks >> 1) * PC_CHUNK + pc_off(base - 1 + tap + r, (ks & 1) * 4 + g);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = *reinterpret_cast<const float4*>(p + j * 2048);
  };
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < RING; ++s) wload(s, s);
  float4 xa[8], xb[8];
  xread(xa, 0);
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    float4 (&cur)[8] = (s & 1) ? xb : xa;
    float4 (&nxt)[8] = (s & 1) ? xa : xb;
    if (s + 1 < STEPS) xread(nxt, s + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const f32x4 w = wr[s % RING][i];
#pragma unroll
      for (int j = 0; j < 8; ++j) PC_MMA(w, cur[j], acc[i][j])
    }
    if (s + RING < STEPS) wload(s % RING, s + RING);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ uint2 pc_pack4(float a, float b, float c, float d) {
  bf16x4 h;
  h[0] = (dx_h16)a; h[1] = (dx_h16)b; h[2] = (dx_h16)c; h[3] = (dx_h16)d;
  return __builtin_bit_cast(uint2, h);
}

// Workgroup -> tile list.  Tiles of utterance b: ceil(len_b / 122) (only tokens < len matter to the loss and to the mel gradient).
// pre[b] = tiles before utterance b, pre[B] = all; written by wave 0, read after a barrier.
__device__ __forceinline__ void pc_tile_prefix(const int* __restrict__ lens, int B, int T, int* pre, int tid) {
  if (tid < 64) {
    int run = 0;
    for (int base = 0; base < B; base += 64) {
      const int i = base + tid;
      const int cnt = i < B ? (min(max(lens[i], 0), T) + PC_TOK - 1) / PC_TOK : 0;
      int inc = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (tid >= off) inc += v; }
      if (i < B) pre[i] = run + inc - cnt;
      run += __shfl(inc, 63, 64);
    }
    if (tid == 0) pre[B] = run;
  }
}
__device__ __forceinline__ void pc_find_tile(const int* pre, int B, int tile, int& b, int& n0) {
  int lo = 0, hi = B;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= tile) lo = mid; else hi = mid; }
  b = lo;
  n0 = (tile - pre[lo]) * PC_TOK;
}

// ---- forward ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void pitch_fwd_kernel(const PitchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const imgA = smem;
  unsigned char* const imgB = smem + PC_IMG;
  unsigned* const mask_s = reinterpret_cast<unsigned*>(smem + 2 * PC_IMG);
  int* const pre = reinterpret_cast<int*>(smem + 2 * PC_IMG + PC_MASK);            // [B + 1] (B <= PC_MAX_B)
  float* const red = reinterpret_cast<float*>(smem + 2 * PC_IMG + PC_MASK + PC_PRE);        // [4][128] partial sums of the last layer
  float* const w3s = reinterpret_cast<float*>(smem + 2 * PC_IMG + PC_MASK + PC_PRE + PC_RED);   // [256][3]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  pc_tile_prefix(a.lens, a.B, a.T, pre, tid);
  for (int u = tid; u < 768; u += 512) w3s[u] = a.w3[u];
  __syncthreads();
  const int ntiles = pre[a.B];
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b, n0;
    pc_find_tile(pre, a.B, tile, b, n0);
    b = __builtin_amdgcn_readfirstlane(b); n0 = __builtin_amdgcn_readfirstlane(n0);
    __syncthreads();                                   // the previous tile's readers of the images / scratch are done
    // ---- window of the mel -> image A (channels 0..79 real, 80..95 zero), sign-bit staging area zeroed -----------------------------------
    for (int u = tid; u < 96 * PC_IR; u += 512) {      // the first layer walks 96 input channels (three K steps): channels M..95 are zeros
      const int c = u / PC_IR, p = u - c * PC_IR;
      const int n = n0 - 4 + p;
      const float v = (c < a.M && n >= 0 && n < a.T) ? a.mel[((size_t)b * a.M + c) * a.T + n] : 0.f;
      *reinterpret_cast<dx_h16*>(imgA + (c >> 6) * PC_CHUNK + pc_off(p, (c & 63) >> 3) + (c & 7) * 2) = (dx_h16)v;
    }
    for (int u = tid; u < PC_MASK / 4; u += 512) mask_s[u] = 0u;
    __syncthreads();

    // ---- the three MFMA layers ----------------------------------------------------------------------------------------------------------
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
      const unsigned char* in = (l & 1) ? imgB : imgA;
      unsigned char* out = (l & 1) ? imgA : imgB;
      const int base = l + 1;
      f32x4 acc[2][8];
      if (l == 0) pc_mfma_pass<3, 2>(a.w0, 16, 4, wave * 2, in, base, lane, acc);          // pack (256 rows, K 80 -> 128): 96 of the 128 are walked
      else pc_mfma_pass<8, 2>(l == 1 ? a.w1 : a.w2, 16, 8, wave * 2, in, base, lane, acc);
      const float* bias = l == 0 ? a.b0 : (l == 1 ? a.b1 : a.b2);
      const float* sc = l == 0 ? a.s0 : (l == 1 ? a.s1 : a.s2);
      const float* sh = l == 0 ? a.t0 : (l == 1 ? a.t1 : a.t2);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch0 = wave * 32 + i * 16 + 4 * g;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + ch0), sv = *reinterpret_cast<const f32x4*>(sc + ch0), tv = *reinterpret_cast<const f32x4*>(sh + ch0);
        unsigned char* const obase = out + (wave >> 1) * PC_CHUNK + ((g & 1) << 3);
        const int slot = (wave & 1) * 4 + i * 2 + (g >> 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = base + 16 * j + r, n = n0 - 4 + row;
          const bool exists = n >= 0 && n < a.T;
          float x[4];
          unsigned bits = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[i][j][e] + bv[e];
            bits |= (v > 0.f ? 1u : 0u) << e;
            x[e] = exists ? fmaxf(v, 0.f) * sv[e] + tv[e] : 0.f;
          }
          *reinterpret_cast<uint2*>(obase + pc_off(row, slot)) = pc_pack4(x[0], x[1], x[2], x[3]);
          if (exists && row <= 128 - l && bits) atomicOr(&mask_s[(l * PC_IR + row) * 8 + wave], bits << (i * 16 + 4 * g));
        }
      }
      __syncthreads();
    }
    // ---- sign bits -> HBM (valid rows of each layer: l + 1 .. 128 - l), the last layer by dot products ----------------------------------------
    for (int u = tid; u < 3 * PC_IR * 8; u += 512) {
      const int d = u & 7, row = (u >> 3) % PC_IR, l = u / (8 * PC_IR);
      const int n = n0 - 4 + row;
      if (row >= l + 1 && row <= 128 - l && n >= 0 && n < a.T) a.masks[(((size_t)b * a.T + n) * 3 + l) * 8 + d] = mask_s[u];
    }
    {
      const int p = tid & 127, cg = __builtin_amdgcn_readfirstlane(tid >> 7);      // output row 4 + p, channels 64 cg .. 64 cg + 63 (image B holds layer 2's output)
      float sum = 0.f;
      const unsigned char* const ib = imgB + cg * PC_CHUNK;
#pragma unroll
      for (int tap = 0; tap < 3; ++tap) {
        const int row = 3 + p + tap;                                                // rows up to 132 are read by p >= 122 (discarded below): clamp
        const int rr = min(row, PC_IR - 1);
#pragma unroll
        for (int slot = 0; slot < 8; ++slot) {
          const bf16x8 xv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(ib + pc_off(rr, slot)));
#pragma unroll
          for (int e = 0; e < 8; ++e) sum += (float)xv[e] * w3s[(cg * 64 + slot * 8 + e) * 3 + tap];
        }
      }
      red[cg * 128 + p] = sum;
      __syncthreads();
      if (tid < PC_TOK) {
        const int n = n0 + tid;
        if (n < min(a.lens[b], a.T)) a.pp[(size_t)b * a.T + n] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid] + a.b3;
      }
    }
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void w3s_init(unsigned char* smem, const float* w3, int u) {
  reinterpret_cast<float*>(smem + 2 * PC_IMG + PC_MASK + PC_PRE + PC_RED)[u] = w3[u];
}
__global__ __launch_bounds__(512) void pitch_bwd_kernel(const PitchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const imgA = smem;
  unsigned char* const imgB = smem + PC_IMG;
  unsigned* const mask_s = reinterpret_cast<unsigned*>(smem + 2 * PC_IMG);
  int* const pre = reinterpret_cast<int*>(smem + 2 * PC_IMG + PC_MASK);
  float* const grow = reinterpret_cast<float*>(smem + 2 * PC_IMG + PC_MASK + PC_PRE);       // [PC_IR + 2] the window of dpp
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  pc_tile_prefix(a.lens, a.B, a.T, pre, tid);
  for (int u = tid; u < 768; u += 512) w3s_init(smem, a.w3, u);
  if (tid < 256) reinterpret_cast<float*>(smem + 2 * PC_IMG + PC_MASK + PC_PRE)[256 + tid] = a.s2[tid];
  __syncthreads();
  const int ntiles = pre[a.B];
  // the VALU layer's constants (the last conv's weights, the scale of the stage below it) wait in LDS: held in registers across the tile loop they
  // spilled 28 registers, and a scratch reload between the MFMA passes' weight loads drains their queue
  float* const w3s = reinterpret_cast<float*>(smem + 2 * PC_IMG + PC_MASK + PC_PRE + PC_RED);   // [256][3]
  float* const s2s = grow + 256;                                                                 // [256]
  const int vslot = tid & 31;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b, n0;
    pc_find_tile(pre, a.B, tile, b, n0);
    b = __builtin_amdgcn_readfirstlane(b); n0 = __builtin_amdgcn_readfirstlane(n0);
    __syncthreads();
    // ---- window of dpp and of the sign bits ----------------------------------------------------------------------------------------------
    for (int p = tid; p < PC_IR + 2; p += 512) {
      const int n = n0 - 4 + p;
      grow[p] = (n >= 0 && n < a.T) ? a.dpp[(size_t)b * a.T + n] : 0.f;
    }
    for (int u = tid; u < 3 * PC_IR * 8; u += 512) {
      const int d = u & 7, row = (u >> 3) % PC_IR, l = u / (8 * PC_IR);
      const int n = n0 - 4 + row;
      mask_s[u] = (n >= 0 && n < a.T) ? a.masks[(((size_t)b * a.T + n) * 3 + l) * 8 + d] : 0u;
    }
    __syncthreads();
    // ---- d(x2)[m][c] = sum_t w3[c][t] g[m - t + 1]; x scale2 x mask2 -> image A, rows 1 .. 128 ----------------------------------------------------
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int row = 1 + (tid >> 5) + 16 * k, n = n0 - 4 + row;
      const bool exists = n >= 0 && n < a.T;
      const float g0 = grow[row + 1], g1 = grow[row], g2 = grow[row - 1];         // taps 0, 1, 2 meet g[m + 1], g[m], g[m - 1]
      const unsigned mb = (mask_s[(2 * PC_IR + row) * 8 + (vslot >> 2)] >> ((vslot & 3) * 8)) & 0xffu;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* w = w3s + (vslot * 8 + e) * 3;
        const float dx = w[0] * g0 + w[1] * g1 + w[2] * g2;
        v[e] = (exists && ((mb >> e) & 1u)) ? dx * s2s[vslot * 8 + e] : 0.f;
      }
      bf16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (dx_h16)v[e];
      *reinterpret_cast<f32x4*>(imgA + (vslot >> 3) * PC_CHUNK + pc_off(row, vslot & 7)) = __builtin_bit_cast(f32x4, h);
    }
    __syncthreads();
    // ---- conv^T(W2) -> x scale1 x mask1 -> image B (rows 2 ..), conv^T(W1) -> x scale0 x mask0 -> image A (rows 3 ..) ------------------------------
#pragma unroll 1
    for (int l = 0; l < 2; ++l) {
      const unsigned char* in = l == 0 ? imgA : imgB;
      unsigned char* out = l == 0 ? imgB : imgA;
      const int base = l + 2, ml = 1 - l;                                          // the stage whose scale / mask applies: 1, then 0
      f32x4 acc[2][8];
      pc_mfma_pass<8, 2>(l == 0 ? a.w2 : a.w1, 16, 8, wave * 2, in, base, lane, acc);
      const float* sc = ml == 1 ? a.s1 : a.s0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ch0 = wave * 32 + i * 16 + 4 * g;
        const f32x4 sv = *reinterpret_cast<const f32x4*>(sc + ch0);
        unsigned char* const obase = out + (wave >> 1) * PC_CHUNK + ((g & 1) << 3);
        const int slot = (wave & 1) * 4 + i * 2 + (g >> 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = base + 16 * j + r, n = n0 - 4 + row;
          const bool exists = n >= 0 && n < a.T && row < PC_IR;
          const unsigned mb = exists ? (mask_s[(ml * PC_IR + row) * 8 + wave] >> (i * 16 + 4 * g)) & 0xfu : 0u;
          float x[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = ((mb >> e) & 1u) ? acc[i][j][e] * sv[e] : 0.f;
          if (row < PC_IR) *reinterpret_cast<uint2*>(obase + pc_off(row, slot)) = pc_pack4(x[0], x[1], x[2], x[3]);
        }
      }
      __syncthreads();
    }
    // ---- conv^T(W0): 80 output channels = five row blocks (waves 0 .. 4), rows 4 .. 125 -> dmel (B, M, T) += ------------------------------------------
    if (wave * 16 < a.M) {
      f32x4 acc[1][8];
      pc_mfma_pass<8, 1>(a.w0, 8, 8, wave, imgA, 4, lane, acc);                    // backward pack of layer 0: rows 80 -> 128, K 256
      const int ch0 = wave * 16 + 4 * g;
      const int len_b = min(a.lens[b], a.T);           // tokens beyond the length: the model masks their gradient, nothing is written there
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int p = 16 * j + r, n = n0 + p;
        if (p < PC_TOK && n < len_b) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ch0 + e < a.M) a.dmel[((size_t)b * a.M + ch0 + e) * a.T + n] += acc[0][j][e];
        }
      }
    }
  }
}

}  // namespace

extern "C" {

// pp (B, T) = the frozen predictor on mel (B, 80, T); masks (B, T, 3, 8) uint32 for dx_pitch_chain_bwd.  w0..w2: forward packs of the three
// 256-wide convolutions (dx_pack_weights, 16-bit), b*: biases, s* / t*: folded BatchNorm scale / shift, w3: row 0 of the last conv's weight in
// checkpoint layout (256, 3) fp32, b3: its bias.  Replaces loss.py:131-136 applied through dx_conv_gemm / dx_channel_affine / dx_transpose.
int dx_pitch_chain_fwd(const float* mel, int B, int M, int T, const int* lens, const void* w0, const void* w1, const void* w2,
                       const float* b0, const float* b1, const float* b2, const float* s0, const float* s1, const float* s2,
                       const float* t0, const float* t1, const float* t2, const float* w3, float b3, float* pp, void* masks, void* stream) {
  DX_REQUIRE(mel && lens && w0 && w1 && w2 && b0 && b1 && b2 && s0 && s1 && s2 && t0 && t1 && t2 && w3 && pp && masks, "dx_pitch_chain_fwd: null pointer");
  DX_REQUIRE(B > 0 && B <= PC_MAX_B && T > 0 && M > 64 && M <= 96, "dx_pitch_chain_fwd: B <= %d and 64 < n_mel <= 96 (the first layer's pack must be 128 wide: B=%d M=%d)", PC_MAX_B, B, M);
  PitchArgs a{};
  a.mel = mel; a.B = B; a.M = M; a.T = T; a.lens = lens;
  a.w0 = (const dx_h16*)w0; a.w1 = (const dx_h16*)w1; a.w2 = (const dx_h16*)w2;
  a.b0 = b0; a.b1 = b1; a.b2 = b2; a.s0 = s0; a.s1 = s1; a.s2 = s2; a.t0 = t0; a.t1 = t1; a.t2 = t2;
  a.w3 = w3; a.b3 = b3; a.pp = pp; a.masks = (unsigned*)masks;
  static bool configured = false;
  if (!configured) {
    DX_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&pitch_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PC_SMEM) == hipSuccess,
               "dx_pitch_chain_fwd: cannot reserve %d bytes of LDS", PC_SMEM);
    configured = true;
  }
  const int grid = std::min(256, B * dx_cdiv(T, PC_TOK));
  hipLaunchKernelGGL(pitch_fwd_kernel, dim3(grid), dim3(512), PC_SMEM, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK("dx_pitch_chain_fwd");
  return DX_OK;
}

// dmel (B, 80, T) += d(loss)/d(mel) through the frozen predictor, from dpp (B, T) and the forward's masks.  w0..w2: the BACKWARD packs of the
// three 256-wide convolutions.  Replaces loss.py's input-gradient chain (four dx_conv_gemm(transpose) launches + dx_transpose(add)).
int dx_pitch_chain_bwd(const float* dpp, int B, int M, int T, const int* lens, const void* w0, const void* w1, const void* w2,
                       const float* s0, const float* s1, const float* s2, const float* w3, const void* masks, float* dmel, void* stream) {
  DX_REQUIRE(dpp && lens && w0 && w1 && w2 && s0 && s1 && s2 && w3 && masks && dmel, "dx_pitch_chain_bwd: null pointer");
  DX_REQUIRE(B > 0 && B <= PC_MAX_B && T > 0 && M > 64 && M <= 96, "dx_pitch_chain_bwd: B <= %d and 64 < n_mel <= 96 (B=%d M=%d)", PC_MAX_B, B, M);
  PitchArgs a{};
  a.dpp = dpp; a.B = B; a.M = M; a.T = T; a.lens = lens; a.dmel = dmel;
  a.w0 = (const dx_h16*)w0; a.w1 = (const dx_h16*)w1; a.w2 = (const dx_h16*)w2;
  a.s0 = s0; a.s1 = s1; a.s2 = s2; a.w3 = w3; a.masks = (unsigned*)const_cast<void*>(masks);
  static bool configured = false;
  if (!configured) {
    DX_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&pitch_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PC_SMEM) == hipSuccess,
               "dx_pitch_chain_bwd: cannot reserve %d bytes of LDS", PC_SMEM);
    configured = true;
  }
  const int grid = std::min(256, B * dx_cdiv(T, PC_TOK));
  hipLaunchKernelGGL(pitch_bwd_kernel, dim3(grid), dim3(512), PC_SMEM, (hipStream_t)stream, a);
  DX_LAUNCH_CHECK("dx_pitch_chain_bwd");
  return DX_OK;
}

}  // extern "C"

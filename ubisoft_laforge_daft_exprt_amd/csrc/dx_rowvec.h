// Row-pass helpers shared by the LayerNorm kernels (dx_rows.hip) and the fused attention + out-projection + LayerNorm launch (dx_attention.hip).
// Included inside each file's anonymous namespace.
#pragma once
// Row layout in a wave: LPR lanes share one row and a wave walks 64/LPR rows at once.  C = 128 rows are only 512 bytes,
// so one row per wave-instruction leaves the memory pipe mostly idle; 16 lanes x two 16-byte accesses per row puts
// four rows (2 KB per tensor) in flight per wave.  C = 1024 keeps the whole wave on one row.
template <int C> struct RowVec {
  static constexpr int V = 4;                           // floats per access
  static constexpr int LPR = (C >= 256) ? 64 : 16;      // lanes per row
  static constexpr int RPW = 64 / LPR;                  // rows per wave
  static constexpr int K = C / (LPR * V);               // accesses per lane
  static constexpr int E = K * V;
  static_assert(C % (LPR * V) == 0, "C must be a multiple of 64");
};

// p = base of the lane's row, l = lane % LPR
template <int C>
__device__ __forceinline__ void row_load(const float* p, int l, float v[RowVec<C>::E]) {
#pragma unroll
  for (int k = 0; k < RowVec<C>::K; ++k) {
    const float4 t = *reinterpret_cast<const float4*>(p + (k * RowVec<C>::LPR + l) * 4);
    v[k * 4 + 0] = t.x; v[k * 4 + 1] = t.y; v[k * 4 + 2] = t.z; v[k * 4 + 3] = t.w;
  }
}
template <int C>
__device__ __forceinline__ void row_store(float* p, int l, const float v[RowVec<C>::E]) {
#pragma unroll
  for (int k = 0; k < RowVec<C>::K; ++k)
    *reinterpret_cast<float4*>(p + (k * RowVec<C>::LPR + l) * 4) = make_float4(v[k * 4], v[k * 4 + 1], v[k * 4 + 2], v[k * 4 + 3]);
}
// bf16-stored rows: same lane -> column map, 8-byte accesses
template <int C>
__device__ __forceinline__ void row_load(const dx_h16* p, int l, float v[RowVec<C>::E]) {
#pragma unroll
  for (int k = 0; k < RowVec<C>::K; ++k) {
    const bf16x4 t = *reinterpret_cast<const bf16x4*>(p + (k * RowVec<C>::LPR + l) * 4);
    v[k * 4 + 0] = (float)t[0]; v[k * 4 + 1] = (float)t[1]; v[k * 4 + 2] = (float)t[2]; v[k * 4 + 3] = (float)t[3];
  }
}
template <int C>
__device__ __forceinline__ void row_store(dx_h16* p, int l, const float v[RowVec<C>::E]) {
#pragma unroll
  for (int k = 0; k < RowVec<C>::K; ++k) {
    bf16x4 t;
    t[0] = (dx_h16)v[k * 4]; t[1] = (dx_h16)v[k * 4 + 1]; t[2] = (dx_h16)v[k * 4 + 2]; t[3] = (dx_h16)v[k * 4 + 3];
    *reinterpret_cast<bf16x4*>(p + (k * RowVec<C>::LPR + l) * 4) = t;
  }
}
template <int C> __device__ __forceinline__ int row_col(int l, int idx) {
  return ((idx / 4) * RowVec<C>::LPR + l) * 4 + (idx % 4);
}
// sum over the LPR lanes that share a row (every lane of the group gets the total)
template <int C> __device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int off = RowVec<C>::LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// keep/scale factors of the lane's E elements of a row: its K groups of 4 consecutive channels are one 64-bit draw each (the same
// values dx_dropout_scale() gives element by element, at a quarter of the hashing)
template <int C>
__device__ __forceinline__ void row_dropout(uint64_t seed, uint64_t row, int l, uint32_t thresh, float inv_keep, float f[RowVec<C>::E]) {
#pragma unroll
  for (int k = 0; k < RowVec<C>::K; ++k) dx_dropout_scale4(seed, row * C + (uint64_t)((k * RowVec<C>::LPR + l) * 4), thresh, inv_keep, f + 4 * k);
}

